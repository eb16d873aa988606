"""Register / scratch / LDS budget per kernel from the device assembly's metadata (hipcc -S --cuda-device-only)."""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
