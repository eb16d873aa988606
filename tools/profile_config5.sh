#!/bin/bash
# the config-5 shard alone (the other workloads' kernels did not change): kernel stats of the bench command + both counter passes
set -o pipefail
RN=${1:-6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r$RN
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
n=config5_shard
rocprofv3 --output-format csv --kernel-trace --stats -d $O/$n -o $n -- python3 $R/bench.py --config configs4_shard --steps 100 --warmup 10 --no-cpu-baseline > $O/$n.json 2> $O/$n.err || { echo FAILED $n; tail -5 $O/$n.err; exit 1; }
head -8 $O/$n/${n}_kernel_stats.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -size +3M -delete; find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cd $R && bash tools/pmc_config5_mfma.sh > /dev/null && bash tools/pmc_config5_traffic.sh > /dev/null
head -4 gpurun_out/pmc_c5m/summary.txt; head -12 gpurun_out/pmc_c5/traffic.txt
