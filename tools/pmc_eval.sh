#!/bin/bash
# counters of the forward-only evaluation's launches (evalf_rows: csrc/evalf.hpp), one --pmc pass each, summarised by tools/pmc_clock.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_eval
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/ev_fwd.py <<PY
import sys, os
sys.path.insert(0, "$R")
import numpy as np, torch
from gmvae_amd.engine import Engine
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((1024, 784)) < 0.87).astype(np.uint8)).cuda()
for _ in range(30):
    e.forward(x, n_samples=50)
torch.cuda.synchronize()
PY
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set --kernel-trace -d $O/run$i -o pmc -- python3 /tmp/ev_fwd.py > $O/run$i.log 2> $O/run$i.err || { echo FAILED $i; tail -5 $O/run$i.err; continue; }
  python3 $R/tools/pmc_clock.py $O/run$i | grep -A3 "evalf_rows" | head -8
  python3 $R/tools/pmc_summary.py $(find $O/run$i -name "*counter_collection.csv") | grep -A1 "evalf_rows"
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
