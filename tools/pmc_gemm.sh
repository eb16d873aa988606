set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_gemm
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace -d $O/run -o pmc -- python3 $R/tools/gemm_c5.py 1 > $O/out.txt 2> $O/err.txt || { echo FAILED; tail -5 $O/err.txt; exit 1; }
python3 $R/tools/pmc_clock.py $O/run > $O/summary.txt
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
grep -A1 "gemm_grouped" $O/summary.txt | head -40
