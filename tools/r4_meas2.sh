set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_timed_path.py -m gpu -x -q -s -p no:cacheprovider > gpurun_out/r4_traj2.log 2>&1; echo "traj exit $?"; grep -h "trajectory\]\|passed\|failed\|^E " gpurun_out/r4_traj2.log | tail -12
for c in configs2 configs0; do
  python bench.py --config $c --steps 2000 --warmup 200 --no-cpu-baseline --no-iwae-bound --levels > gpurun_out/r4_c_$c.json 2> gpurun_out/r4_c_$c.err
  echo "== $c: $(python -c "import json;j=json.load(open('gpurun_out/r4_c_$c.json'));print(round(j['ms_per_step']*1e3,2),'us/step', j['roofline']['schedule'], j['roofline']['levels'])")"
done
GMVAE_STAMPS=1 python tools/stamps.py > gpurun_out/r4_stamps1b.log 2>&1; tail -6 gpurun_out/r4_stamps1b.log
