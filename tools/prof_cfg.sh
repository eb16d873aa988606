#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of one bench config; prints the top kernels.   usage: prof_cfg.sh <config> [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_cfg
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $O/$1 -o $1 -- python3 $R/bench.py --config $1 --steps ${2:-400} --warmup 40 --no-cpu-baseline --no-iwae-bound > $O/$1.json 2> $O/$1.err || { echo FAILED; tail -5 $O/$1.err; exit 1; }
python3 - $O/$1/$1_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {r['Percentage']} %")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
