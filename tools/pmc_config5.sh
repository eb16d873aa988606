set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_c5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $O/run -o pmc -- python3 $R/bench.py --config configs4_shard --steps 20 --warmup 4 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo FAILED; tail -5 $O/*.err; exit 1; }
python3 $R/tools/pmc_clock.py $O/run > $O/summary.txt
# fabric traffic of the same launches (FETCH_SIZE and WRITE_SIZE in passes of their own)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $c --kernel-trace -d $O/$c -o pmc -- python3 $R/bench.py --config configs4_shard --steps 10 --warmup 2 --no-cpu-baseline > $O/$c.json 2> $O/$c.err || { echo FAILED $c; tail -5 $O/$c.err; exit 1; }
done
python3 $R/tools/traffic_from_pmc.py $O/FETCH_SIZE/pmc_counter_collection.csv $O/WRITE_SIZE/pmc_counter_collection.csv $O/traffic.json > $O/traffic.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cat $O/summary.txt
