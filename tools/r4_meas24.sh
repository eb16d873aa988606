cd $GRAFT_REPO_ROOT
run() {
  echo "== $1 $2"
  env $2 timeout -k 10 120 python bench.py --config $1 --steps 400 --warmup 40 --no-cpu-baseline --levels > gpurun_out/r4_sk_tmp.log 2>&1; grep "  sk_" gpurun_out/r4_sk_tmp.log | awk '{printf "%s %s | ", $1, $6}'; echo; grep "^{" gpurun_out/r4_sk_tmp.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step']*1e3, 'us', j['parity']['rel_err'])"
}
(
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_timed_path.py -m gpu -q -p no:cacheprovider -x -k "skinny or H512 or trajectory" 2>&1 | tail -3
for w in 8 6 5 4; do run run_train "GMVAE_SK_WPW=$w"; done
run run_train "A=1"
) > gpurun_out/r4_meas24.log 2>&1
cat gpurun_out/r4_meas24.log
