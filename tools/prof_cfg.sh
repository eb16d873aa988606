#!/bin/bash
# rocprofv3 --kernel-trace --stats of `bench.py --config $1` (on the GPU box): prints the per-kernel table, keeps the stats csv
# under gpurun_out/prof_$1/.  usage: bash tools/prof_cfg.sh <config> [steps]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $O -o p -- python3 $R/bench.py --config $1 --steps ${2:-1000} --warmup 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { echo FAILED; tail -5 $O/bench.err; exit 1; }
rm -f $O/*kernel_trace.csv $O/*.db $O/*agent_info*
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/p_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])*1e-3:8.2f} us x {r['Calls']:>7s}  {r['Name'][:90]}")
PY
python3 -c "import json;j=json.load(open('$O/bench.json'));print('ms_per_step', j['ms_per_step'])"
