#!/bin/bash
# interleaved comparison of several builds of the library on one box: tools/ab/lib_<tag>.so for each tag given
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cp gmvae_amd/lib/libgmvae_hip.so /tmp/keep.so
CFG=${CFG:-configs4_shard}; STEPS=${STEPS:-50}
for i in 1 2 3; do
  for t in "$@"; do
    cp tools/ab/lib_$t.so gmvae_amd/lib/libgmvae_hip.so
    python bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(j['ms_per_step']*1e3,1), 'us')"
  done
done
cp /tmp/keep.so gmvae_amd/lib/libgmvae_hip.so
