# fabric traffic of the config-5 shard's launches (FETCH_SIZE and WRITE_SIZE in passes of their own, kernel-trace only)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_c5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $c --kernel-trace -d $O/$c -o pmc -- python3 $R/tools/pmc_step.py 512 512 6 0 64 3072 64 50 > $O/$c.json 2> $O/$c.err || { echo FAILED $c; tail -5 $O/$c.err; exit 1; }
done
python3 $R/tools/traffic_from_pmc.py $O/FETCH_SIZE/pmc_counter_collection.csv $O/WRITE_SIZE/pmc_counter_collection.csv $O/traffic.json > $O/traffic.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cat $O/traffic.txt
