cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_timed_path.py tests/test_model_api.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r4_sk_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/r4_sk_tests.log
run() {
  echo "== $1 $2"
  env $2 timeout -k 10 120 python bench.py --config $1 --steps 200 --warmup 20 --no-cpu-baseline --levels > gpurun_out/r4_sk_tmp.log 2>&1; grep "  sk_" gpurun_out/r4_sk_tmp.log | awk '{printf "%s %s | ", $1, $6}'; echo; grep "^{" gpurun_out/r4_sk_tmp.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step']*1e3, 'us')"
}
for cfg in configs2_h512 configs1_h512; do
  run $cfg "GMVAE_SK_WIDE=0 GMVAE_SK_TM=4"
  run $cfg "GMVAE_SK_WIDE=1 GMVAE_SK_TM=4"
  run $cfg "GMVAE_SK_WIDE=1 GMVAE_SK_TM=2"
  run $cfg "GMVAE_SK_WIDE=1 GMVAE_SK_TM=2 GMVAE_SK_RT=2"
  run $cfg "GMVAE_SK_WIDE=1 GMVAE_SK_TM=2 GMVAE_SK_RT=1"
  run $cfg "A=1"
done
run run_train "A=1"
python tools/skstamps.py 1024 64 F5,B1,W
