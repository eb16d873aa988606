"""profiles/round1_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes).  gfx950 correction: FETCH_SIZE counts 128-byte requests at 64 bytes for wide
coalesced streams (16 B/lane loads and LDS-DMA), so read bytes = 2 * FETCH_SIZE KB; WRITE_SIZE is exact."""
import csv, sys, json, collections
fetch, write = collections.defaultdict(list), collections.defaultdict(list)
for path, dst, name in ((sys.argv[1], fetch, "FETCH_SIZE"), (sys.argv[2], write, "WRITE_SIZE")):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            dst[(r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
out = {}
for key in fetch:
    f = sum(fetch[key]) / len(fetch[key]); w = sum(write.get(key, [0])) / max(len(write.get(key, [0])), 1)
    out[f"{key[0]} grid={key[1]}"] = {"fetch_kb_raw": f, "write_kb": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "launches": len(fetch[key])}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if "gmvae" in k: print(f"{k:80s} {v['hbm_bytes_per_launch']/1e6:8.2f} MB/launch")
