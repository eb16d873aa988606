#!/bin/bash
# On the GPU box (through gpurun): the skinny schedule's parity tests, then the per-launch table of the named bench configs
# (timeline share per launch, us) and their step time.   usage: sk_levels.sh [config ...]   (default: configs2_h512 configs1_h512 run_train)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_timed_path.py -m gpu -q -p no:cacheprovider -x -k "skinny or H512 or trajectory" 2>&1 | tail -3
for cfg in ${@:-configs2_h512 configs1_h512 run_train}; do
  echo "== $cfg"
  timeout -k 10 120 python bench.py --config $cfg --steps 400 --warmup 40 --no-cpu-baseline --levels > gpurun_out/sk_levels_$cfg.log 2>&1
  grep "  sk_" gpurun_out/sk_levels_$cfg.log | awk '{printf "%s %s | ", $1, $6}'; echo
  grep "^{" gpurun_out/sk_levels_$cfg.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step']*1e3, 'us per step; ELBO rel err', j['parity']['rel_err'])"
done
