# matrix-pipe / LDS / wait counters of the skinny schedule's launches at hidden 512, batch 1024 (one --pmc pass, kernel-trace only)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_h512
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $O/run -o pmc -- python3 $R/tools/pmc_step.py 512 ${1:-1024} 12 16 ${2:-64} > $O/run.log 2> $O/run.err || { echo FAILED; tail -5 $O/run.err; exit 1; }
python3 $R/tools/pmc_clock.py $O/run > $O/summary.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cat $O/summary.txt
