cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --config configs4_shard --steps 30 --warmup 5 --no-cpu-baseline --levels > gpurun_out/r4_c5.log 2>&1
grep -v "^{" gpurun_out/r4_c5.log | tail -40
grep "^{" gpurun_out/r4_c5.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step']*1e3, 'us', j['config']['workload']); print(j['roofline'])"
