"""The config-5 shard's large GEMM shapes through gmvae_gemm_test (cfg 2 = the 128x128x32 instance): TFLOP/s per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
shapes = [("fwd_dec  NN 25600x3072x512", 25600, 3072, 512, 0), ("bwd_dX   NT 25600x512x3072", 25600, 512, 3072, 1),
          ("bwd_dW   TN 512x3072x25600", 512, 3072, 25600, 2), ("hidden   NN 25600x512x512", 25600, 512, 512, 0),
          ("square   NN 8192x4096x4096", 8192, 4096, 4096, 0)]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    for name, M, N, K, tr in shapes:
        us, tf = bench(M, N, K, tr, False, 2, 1, iters=30)
        print(f"{name:30s}: {us:9.2f} us  {tf:8.2f} TFLOP/s", flush=True)
