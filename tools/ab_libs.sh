#!/bin/bash
# A/B of library builds on one box, interleaved: tools/ab_libs.sh <rounds> <config> lib1.so lib2.so ... ("default" = the in-tree build)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
rounds=$1; cfg=$2; shift 2
for i in $(seq $rounds); do
  for l in "$@"; do
    if [ $l = default ]; then unset GMVAE_HIP_LIB; else export GMVAE_HIP_LIB=$PWD/$l; fi
    python tools/step_time.py $cfg 1.5 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', round(j['us_per_step_median'],2), 'us/step  min', round(j['us_per_step_min'],2), 'levels', j['levels'], 'timeouts', j['timeouts'])"
  done
done
