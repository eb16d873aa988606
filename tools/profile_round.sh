#!/bin/bash
# Round profiles (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats of the default bench line and of the
# other named workloads, then counter passes (each --pmc set in a run of its own, kernel-trace only) over multi-step
# train-graph launches of the headline configuration and of the bin/run_train.sh sizes.   usage: profile_round.sh <round>
set -o pipefail
RN=${1:-5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r$RN
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, bench args...
  local n=$1; shift
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/$n -o $n -- python3 $R/bench.py "$@" > $O/$n.json 2> $O/$n.err || { echo FAILED $n; tail -5 $O/$n.err; exit 1; }
  echo "== $n"; head -6 $O/$n/${n}_kernel_stats.csv | cut -c1-140
}
run bench --no-cpu-baseline
run run_train_sizes --config run_train --steps 2000 --warmup 200 --no-cpu-baseline
run config5_shard --config configs4_shard --steps 100 --warmup 10 --no-cpu-baseline
run configs1 --config configs1 --steps 2000 --warmup 200 --no-cpu-baseline
run configs0 --config configs0 --steps 2000 --warmup 200 --no-cpu-baseline
run configs2_h512 --config configs2_h512 --steps 400 --warmup 40 --no-cpu-baseline --no-iwae-bound
run configs1_h512 --config configs1_h512 --steps 400 --warmup 40 --no-cpu-baseline --no-iwae-bound
run configs3_dp1 --config configs3_dp1 --steps 400 --warmup 40 --no-cpu-baseline --no-iwae-bound
run eval_iwae --config eval_iwae --steps 200 --warmup 20 --no-cpu-baseline
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --output-format csv --pmc $set --kernel-trace -d $O/pmc_$name -o pmc -- python3 $R/tools/pmc_step.py 64 1024 12 16 > $O/pmc_$name.log 2>&1 || { echo FAILED pmc $name; tail -5 $O/pmc_$name.log; exit 1; }
done
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --output-format csv --pmc $set --kernel-trace -d $O/pmcrt_$set -o pmc -- python3 $R/tools/pmc_step.py 512 64 12 16 128 > $O/pmcrt_$set.log 2>&1 || { echo FAILED pmcrt $set; tail -5 $O/pmcrt_$set.log; exit 1; }
done
# keep the summaries only (the merged-back directory is capped at 64 MiB)
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
du -sh $O
