#!/bin/bash
# the thin weight gradients' slab count (GMVAE_NSPLIT_SMALL) against the config-5 shard's step time
for ns in ${NSLIST:-64 48 32 24 16}; do
  GMVAE_NSPLIT_SMALL=$ns python bench.py --config configs4_shard --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ns_small', $ns, round(j['ms_per_step']*1e3,1), 'us')"
done
