set -o pipefail
cd $GRAFT_REPO_ROOT
python tools/traj_diag.py vae 784 2 1 64 100 4 11 12 13 > gpurun_out/r4_trajdiag.log 2>&1; tail -30 gpurun_out/r4_trajdiag.log
for c in configs0 configs1 configs2 configs2_h512 configs1_h512 run_train; do
  python bench.py --config $c --steps 400 --warmup 40 --no-cpu-baseline --no-iwae-bound --levels > gpurun_out/r4_b_$c.json 2> gpurun_out/r4_b_$c.err
  echo "== $c: $(python -c "import json;j=json.load(open('gpurun_out/r4_b_$c.json'));print(round(j['ms_per_step']*1e3,2),'us/step', j['roofline']['schedule'], j['roofline']['levels'])")"
done
GMVAE_NO_SKINNY=1 python bench.py --config configs2_h512 --steps 200 --warmup 20 --no-cpu-baseline --no-iwae-bound --levels > gpurun_out/r4_b_c2h512_general.json 2> gpurun_out/r4_b_c2h512_general.err
echo "== configs2_h512 general: $(python -c "import json;j=json.load(open('gpurun_out/r4_b_c2h512_general.json'));print(round(j['ms_per_step']*1e3,2),'us/step', j['roofline']['schedule'], j['roofline']['levels'])")"
GMVAE_STAMPS=1 python tools/stamps.py > gpurun_out/r4_stamps1.log 2>&1; tail -8 gpurun_out/r4_stamps1.log
GMVAE_STAMPS=4 python tools/stamps.py > gpurun_out/r4_stamps4.log 2>&1; tail -12 gpurun_out/r4_stamps4.log
python tools/dwstamps.py > gpurun_out/r4_dwstamps.log 2>&1; tail -15 gpurun_out/r4_dwstamps.log
