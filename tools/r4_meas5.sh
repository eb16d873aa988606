cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_timed_path.py -m gpu -x -q -s -p no:cacheprovider -k "vae-L2 or vae_gmp-L64-H64" > gpurun_out/r4_m2v.log 2>&1; echo "m2v tests exit $?"; grep -h "trajectory\]\|passed\|failed\|^E " gpurun_out/r4_m2v.log | tail -8
for c in configs0 configs1; do
  timeout -k 10 200 python bench.py --config $c --steps 2000 --warmup 200 --no-cpu-baseline --no-iwae-bound --levels > gpurun_out/r4_e_$c.json 2> gpurun_out/r4_e_$c.err
  echo "== $c: $(python -c "import json;j=json.load(open('gpurun_out/r4_e_$c.json'));print(round(j['ms_per_step']*1e3,2),'us/step', j['roofline']['schedule'], j['roofline']['levels'], j['parity'])")"
done
