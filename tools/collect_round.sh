#!/bin/bash
# gpurun_out/prof_r<N> (scratch, written by tools/profile_round.sh on the GPU box) -> profiles/round<N>_* (tracked)
set -e
RN=${1:-5}
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r$RN
for n in bench run_train_sizes config5_shard configs1 configs0 configs2_h512 configs1_h512 configs3_dp1 eval_iwae; do
  cp $S/$n/${n}_kernel_stats.csv profiles/round${RN}_${n}_kernel_stats.csv
  cp $S/$n.json profiles/round${RN}_${n}_under_rocprof.json
done
python3 tools/pmc_summary.py $S/pmc_FETCH_SIZE/pmc_counter_collection.csv $S/pmc_WRITE_SIZE/pmc_counter_collection.csv \
  $S/pmc_SQ_BUSY_CYCLES/pmc_counter_collection.csv $S/pmc_SQ_WAVE_CYCLES/pmc_counter_collection.csv > profiles/round${RN}_pmc_per_kernel.txt
python3 tools/traffic_from_pmc.py $S/pmc_FETCH_SIZE/pmc_counter_collection.csv $S/pmc_WRITE_SIZE/pmc_counter_collection.csv profiles/round${RN}_traffic.json
python3 tools/traffic_from_pmc.py $S/pmcrt_FETCH_SIZE/pmc_counter_collection.csv $S/pmcrt_WRITE_SIZE/pmc_counter_collection.csv profiles/round${RN}_traffic_run_train_sizes.json
ls -la profiles/round${RN}_*
# (on the GPU box: bash tools/pmc_config5_traffic.sh -> gpurun_out/pmc_c5/traffic.{json,txt}; copied by hand to profiles/round${RN}_traffic_config5_shard.*)
