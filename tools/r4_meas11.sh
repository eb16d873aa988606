cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_timed_path.py tests/test_model_api.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r4_sk_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/r4_sk_tests.log
for cfg in configs2_h512 configs1_h512 run_train; do
  for rt in 1 2 4 d; do
      echo "== $cfg rt=$rt"
      if [ $rt = d ]; then unset GMVAE_SK_RT; else export GMVAE_SK_RT=$rt; fi
      timeout -k 10 120 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --levels > gpurun_out/r4_sk_${cfg}_rt$rt.log 2>&1; grep "  sk_" gpurun_out/r4_sk_${cfg}_rt$rt.log | awk '{printf "%s %s | ", $1, $6}'; echo; grep "^{" gpurun_out/r4_sk_${cfg}_rt$rt.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step']*1e3, 'us')"
  done
done
