#!/bin/bash
# gpurun_out/prof_r2 (scratch, written by tools/profile_round2.sh on the GPU box) -> profiles/round2_* (tracked)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r2
cp $S/bench/bench_kernel_stats.csv profiles/round2_bench_kernel_stats.csv
cp $S/bench.json profiles/round2_bench_under_rocprof.json
cp $S/c5/c5_kernel_stats.csv profiles/round2_config5_shard_kernel_stats.csv
cp $S/c5.json profiles/round2_config5_shard_under_rocprof.json
cp $S/rt/rt_kernel_stats.csv profiles/round2_run_train_sizes_kernel_stats.csv
cp $S/rt.json profiles/round2_run_train_sizes_under_rocprof.json
python3 tools/pmc_summary.py $S/pmc_FETCH_SIZE/pmc_counter_collection.csv $S/pmc_WRITE_SIZE/pmc_counter_collection.csv \
  $S/pmc_SQ_BUSY_CYCLES/pmc_counter_collection.csv $S/pmc_SQ_WAVE_CYCLES/pmc_counter_collection.csv > profiles/round2_pmc_per_kernel.txt
python3 tools/traffic_from_pmc.py $S/pmc_FETCH_SIZE/pmc_counter_collection.csv $S/pmc_WRITE_SIZE/pmc_counter_collection.csv profiles/round2_traffic.json
cp $S/pmc_config5_summary.txt profiles/round2_pmc_config5_clock_mfma.txt
cp $S/pmc_config5_traffic.json profiles/round2_traffic_config5.json
ls -la profiles/round2_*
