#!/bin/bash
# Instruction-fetch counters of the one-launch step (run on the GPU box): which SQ / SQC counters the box lists, then one
# --pmc pass per set over multi-step train-graph launches of the headline configuration.   usage: pmc_ifetch.sh <outdir>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ifetch}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INSTS_WAVE32[A-Z_]*\|SQ_WAVES[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_INST_CYCLES[A-Z_]*\|SQC_TC_INST[A-Z_]*\|SQC_TC_REQ\|SQC_TC_STALL\|SQC_DCACHE_REQ[A-Z_]*\|SQC_DCACHE_HITS\|SQC_DCACHE_MISSES[A-Z_]*" $O/avail.txt | sort -u > $O/names.txt
cat $O/names.txt | tr '\n' ' '; echo
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set --kernel-trace -d $O/p$i -o pmc -- python3 $R/tools/pmc_step.py 64 1024 12 16 > $O/p$i.log 2>&1 || { echo "FAILED set $i: $set"; tail -3 $O/p$i.log; continue; }
  python3 $R/tools/pmc_summary.py $(find $O/p$i -name "*counter_collection.csv") 2>/dev/null | grep -A1 "mega3" | head -4 || true
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
ls $O $O/p1 | head -30
