#!/bin/bash
# interleaved A/B of one environment switch on one box: ab_env.sh VAR [config] [steps]
for i in 1 2 3; do
  for w in off on; do
    if [ $w = on ]; then export $1=1; else unset $1; fi
    python bench.py --config ${2:-configs4_shard} --steps ${3:-50} --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $w', round(j['ms_per_step']*1e3,1), 'us')"
  done
done
