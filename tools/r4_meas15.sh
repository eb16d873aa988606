cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r4_gputest4.log 2>&1; echo "gpu tests exit $?"; tail -3 gpurun_out/r4_gputest4.log
run() {
  echo "== $1 $2"
  env $2 timeout -k 10 120 python bench.py --config $1 --steps 200 --warmup 20 --no-cpu-baseline --levels > gpurun_out/r4_sk_tmp.log 2>&1; grep "  sk_" gpurun_out/r4_sk_tmp.log | awk '{printf "%s %s | ", $1, $6}'; echo; grep "^{" gpurun_out/r4_sk_tmp.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step']*1e3, 'us')"
}
(
for cfg in configs2_h512 configs1_h512 run_train; do
  run $cfg "A=1"
done
) > gpurun_out/r4_meas15.log 2>&1
cat gpurun_out/r4_meas15.log
