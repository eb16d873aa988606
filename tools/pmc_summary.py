"""Summarise rocprofv3 --pmc CSV (counter_collection.csv): mean per (kernel, grid) of each counter."""
import csv, sys, collections, glob
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in d:
    vals = {k: sum(v) / len(v) for k, v in d[key].items()}
    print(key[0], "grid", key[1], "n", len(next(iter(d[key].values()))))
    print("   " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(vals.items())))
