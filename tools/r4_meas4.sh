cd $GRAFT_REPO_ROOT
python -m pytest tests/test_timed_path.py tests/test_hip_parity.py -m gpu -x -q -p no:cacheprovider -k "vae_gmp or gmp or golden" > gpurun_out/r4_gmp.log 2>&1; echo "gmp tests exit $?"; tail -3 gpurun_out/r4_gmp.log
for c in configs1 configs0 configs2; do
  python bench.py --config $c --steps 2000 --warmup 200 --no-cpu-baseline --no-iwae-bound --levels > gpurun_out/r4_d_$c.json 2> gpurun_out/r4_d_$c.err
  echo "== $c: $(python -c "import json;j=json.load(open('gpurun_out/r4_d_$c.json'));print(round(j['ms_per_step']*1e3,2),'us/step', j['roofline']['schedule'], j['roofline']['levels'])")"
done
GMVAE_STAMPS=1 python tools/stamps.py vae_gmp 256 64 10 2>&1 | tail -7
