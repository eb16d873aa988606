cd $GRAFT_REPO_ROOT
# skinny schedule at large batches: first-layer slabs and the W launch's batch split
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_timed_path.py -m gpu -q -p no:cacheprovider -k "skinny or H512 or trajectory" > gpurun_out/r4_sk_tests.log 2>&1; echo "tests exit $?"; tail -3 gpurun_out/r4_sk_tests.log
for cfg in configs2_h512 configs1_h512 run_train; do
  for ks in 1 2 4 8; do
    for ns in 1 4; do
      echo "== $cfg ksplit=$ks ns1=$ns"
      GMVAE_SK_KSPLIT=$ks GMVAE_SK_NS1=$ns timeout -k 10 120 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step']*1e3, 'us')"
    done
  done
done
echo "== default"
for cfg in configs2_h512 configs1_h512 run_train; do timeout -k 10 120 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --levels > gpurun_out/r4_sk_$cfg.log 2>&1; tail -30 gpurun_out/r4_sk_$cfg.log | cut -c1-400; done
