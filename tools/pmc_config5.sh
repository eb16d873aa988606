set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_c5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $O/run -o pmc -- python3 $R/bench.py --config configs4_shard --steps 20 --warmup 4 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo FAILED; tail -5 $O/*.err; exit 1; }
python3 $R/tools/pmc_clock.py $O/run > $O/summary.txt
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cat $O/summary.txt
