#!/bin/bash
# A/B of two builds of the library on one box: tools/ab/libgmvae_hip_old.so (built by hand from an older commit's csrc) against the tree's.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cp gmvae_amd/lib/libgmvae_hip.so /tmp/new.so
for i in 1 2 3; do
  for w in old new; do
    if [ $w = old ]; then cp tools/ab/libgmvae_hip_old.so gmvae_amd/lib/libgmvae_hip.so; else cp /tmp/new.so gmvae_amd/lib/libgmvae_hip.so; fi
    python bench.py --config ${1:-configs4_shard} --steps ${2:-100} --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(j['ms_per_step']*1e3,1), 'us')"
  done
done
cp /tmp/new.so gmvae_amd/lib/libgmvae_hip.so
