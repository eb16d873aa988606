"""Per kernel, from one rocprofv3 --pmc --kernel-trace run (counter_collection.csv [+ kernel_trace.csv]): mean launch
duration, effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X_MICROARCH.md 'DVFS give-back') and the
matrix-pipe utilisation in CYCLES (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles)).

usage: pmc_clock.py <dir with the CSVs>"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
dur = {}
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
per = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = r["Dispatch_Id"]
        k = (r["Kernel_Name"].split("(")[0][:70], r.get("Grid_Size", ""))
        per[k][d][r["Counter_Name"]] = float(r["Counter_Value"])
        if d not in dur and "Start_Timestamp" in r:
            dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for k, disp in sorted(per.items(), key=lambda kv: -sum(dur.get(d, 0) for d in kv[1])):
    n = len(disp)
    ds = [dur[d] for d in disp if d in dur]
    if not ds:
        continue
    mean = lambda c: sum(v.get(c, 0.0) for v in disp.values()) / n
    t = sum(ds) / len(ds)
    cyc = mean("GRBM_GUI_ACTIVE") / 8.0
    line = f"{k[0]} grid {k[1]} n {n}: {t * 1e6:9.1f} us"
    if cyc > 0:
        # GRBM_GUI_ACTIVE also counts the dispatch's set-up and drain: over a launch of a few microseconds "cycles / duration" is not a
        # clock (round 5's table read 4-14 GHz there) -- printed as active cycles only below 50 us
        line += f"  clock {cyc / t * 1e-9:5.2f} GHz" if t >= 50e-6 else f"  (active cycles / 8 XCDs {cyc:.3g}: no clock estimate below 50 us)"
        if mean("SQ_VALU_MFMA_BUSY_CYCLES") > 0:
            line += f"  mfma_busy {mean('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * cyc):6.3f} of cycles"
    print(line)
    print("     " + "  ".join(f"{c}={mean(c):.4g}" for c in sorted(next(iter(disp.values())))))
