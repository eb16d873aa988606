#!/bin/bash
# tuning sweep of the two grouped launches of the fused schedule (P1 split count / tile cfg, dW split count / tile cfg)
for fs in 4 7; do for pc in 0 1; do for ns in 2 4 8; do for dc in 0 1; do
  r=$(GMVAE_FWD_SPLITS=$fs GMVAE_P1_CFG=$pc GMVAE_NSPLIT=$ns GMVAE_DW_CFG=$dc python bench.py --no-cpu-baseline --steps 1500 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "fwd_splits=$fs p1_cfg=$pc nsplit=$ns dw_cfg=$dc ms_per_step=$r"
done; done; done; done
