#!/bin/bash
# the top weight gradient's slab count (GMVAE_NSPLIT_TOP) against the step time at several plane-GEMM shapes
run() {  # batch data_dim hidden nslist
  for ns in $4; do
    GMVAE_NSPLIT_TOP=$ns python bench.py --model gmvae --batch $1 --data-dim $2 --hidden $3 --latent 64 --components 64 --n-samples 50 --steps 30 --warmup 3 --graph-steps 10 --no-cpu-baseline --no-iwae-bound 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B $1 D $2 H $3 ns', $ns, round(j['ms_per_step']*1e3,1), 'us')"
  done
}
run 512 2048 512 "16 10 8 7 6 5 4"
run 256 3072 512 "16 8 6 5 4 3"
run 1024 3072 512 "16 8 6 5 4"
run 512 3072 1024 "16 8 5 4 3 2"
