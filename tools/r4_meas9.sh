cd $GRAFT_REPO_ROOT
for env in "GMVAE_NO_SKINNY=1" "GMVAE_NO_SKINNY=1 GMVAE_PLANES_MINROWS=512" "GMVAE_NO_SKINNY=1 GMVAE_NO_PLANES=1"; do
  echo "== $env"
  env $env timeout -k 10 120 python bench.py --config configs2_h512 --steps 100 --warmup 10 --no-cpu-baseline --levels > gpurun_out/r4_gen_h512.log 2>&1; grep -v "^{" gpurun_out/r4_gen_h512.log | tail -45 | cut -c1-200; grep "^{" gpurun_out/r4_gen_h512.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step']*1e3, 'us', j['config'])"
done
