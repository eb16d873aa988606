#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): kernel-trace stats of the default bench line and of the two
# H = 512 configurations, then counter passes (each --pmc set in a run of its own, kernel-trace only) over a few
# multi-step train-graph launches of the headline configuration.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $O/bench -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo FAILED; tail -5 $O/*.err $O/*.log 2>/dev/null; exit 1; }
rocprofv3 --output-format csv --kernel-trace --stats -d $O/c5 -o c5 -- python3 $R/bench.py --config configs4_shard --steps 100 --warmup 10 --no-cpu-baseline > $O/c5.json 2> $O/c5.err || { echo FAILED; tail -5 $O/*.err $O/*.log 2>/dev/null; exit 1; }
rocprofv3 --output-format csv --kernel-trace --stats -d $O/rt -o rt -- python3 $R/bench.py --config run_train --steps 1000 --warmup 100 --no-cpu-baseline > $O/rt.json 2> $O/rt.err || { echo FAILED; tail -5 $O/*.err $O/*.log 2>/dev/null; exit 1; }
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  rocprofv3 --output-format csv --pmc $set --kernel-trace -d $O/pmc_$name -o pmc -- python3 $R/tools/pmc_step.py 64 1024 12 16 > $O/pmc_$name.log 2>&1 || { echo FAILED; tail -5 $O/*.err $O/*.log 2>/dev/null; exit 1; }
done
# keep the summaries only (the merged-back directory is capped at 64 MiB)
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
du -sh $O; find $O -name "*.csv" | head -40
# config-5 shard: effective clock and matrix-pipe utilisation in cycles per GEMM launch
bash $R/tools/pmc_config5.sh > $O/pmc_config5.log 2>&1 || { echo FAILED pmc_config5; tail -5 $O/pmc_config5.log; }
cp $R/gpurun_out/pmc_c5/summary.txt $O/pmc_config5_summary.txt 2>/dev/null
cp $R/gpurun_out/pmc_c5/traffic.json $O/pmc_config5_traffic.json 2>/dev/null
rm -rf $R/gpurun_out/pmc_c5/run $R/gpurun_out/pmc_c5/FETCH_SIZE $R/gpurun_out/pmc_c5/WRITE_SIZE
du -sh $O
