"""The config-5 shard's mid-size backward GEMM shapes through gmvae_gemm_test, every tile configuration: us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
shapes = [("dz    NT 25600x64x512", 25600, 64, 512, 1, 1), ("dhg   NT 25600x512x128", 25600, 512, 128, 1, 1),
          ("dWd0  TN 64x512x25600", 64, 512, 25600, 2, 16), ("dWg1  TN 512x128x25600", 512, 128, 25600, 2, 16),
          ("dWp   TN 64x128x25600", 64, 128, 25600, 2, 16)]
for name, M, N, K, tr, ns in shapes:
    row = []
    for cfg in (0, 1, 2):
        try:
            us, tf = bench(M, N, K, tr, False, cfg, ns, iters=30)
            row.append(f"cfg{cfg} {us:7.1f} us")
        except Exception as e:
            row.append(f"cfg{cfg} fail")
    print(f"{name:26s}: " + "  ".join(row), flush=True)
