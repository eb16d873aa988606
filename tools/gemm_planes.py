"""The config-5 shard's large GEMM shapes: the fp32 MFMA big-round instance (cfg 2) against the pre-split plane instance
(cfg 5 = planes of the previous cfg-4 call, the GEMM alone; cfg 4 = with the two split_planes launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gemm_bench import bench
shapes = [("fwd_dec  NN 25600x3072x512", 25600, 3072, 512, 0, 1), ("bwd_dX   NT 25600x512x3072", 25600, 512, 3072, 1, 1),
          ("bwd_dW   TN 512x3072x25600", 512, 3072, 25600, 2, 16), ("hidden   NN 25600x512x512", 25600, 512, 512, 0, 1),
          ("square   NN 8192x4096x4096", 8192, 4096, 4096, 0, 1)]
for name, M, N, K, tr, ns in shapes:
    out = []
    for cfg in (2, 4, 5):
        us, tf = bench(M, N, K, tr, False, cfg, ns, iters=20)
        out.append(f"cfg{cfg} {us:8.1f} us {tf:6.1f} TF")
    print(f"{name:30s}: " + " | ".join(out), flush=True)
