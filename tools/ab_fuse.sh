#!/bin/bash
# A/B on one box: mega3_step (one launch per step) against mega2_fwd_bwd + dw_adam (GMVAE_NO_FUSE=1), interleaved.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cfg=${1:-configs2}; rounds=${2:-3}
for i in $(seq $rounds); do
  for w in fused two; do
    if [ $w = two ]; then export GMVAE_NO_FUSE=1; else unset GMVAE_NO_FUSE; fi
    python tools/step_time.py $cfg 1.5 $3 2>&1 | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(j['us_per_step_median'],2), 'us/step  min', round(j['us_per_step_min'],2), 'levels', j['levels'], 'timeouts', j['timeouts'], 'loss', round(j['loss'],4))"
  done
done
