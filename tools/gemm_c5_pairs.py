"""The config-5 shard's three plane GEMMs alone on f16 pairs (cfg 7 of gmvae_gemm_test: operands split once by a cfg-6 call): us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gmvae_amd import _lib as L
def t(M, N, K, tr, ns, iters=20):
    A = torch.randn(K, M, device="cuda") if tr == 2 else torch.randn(M, K, device="cuda")
    W = torch.randn(K, N, device="cuda") if tr != 1 else torch.randn(N, K, device="cuda")
    C = torch.empty(ns, M + 1, N, device="cuda") if tr == 2 else torch.empty(M, N, device="cuda")
    def run(cfg):
        L.check(L.lib.gmvae_gemm_test(L.ptr(A), 0, L.ptr(W), L.ptr(W) if tr == 2 else None, L.ptr(C), M, N, K, tr, 0, cfg, ns, L.current_stream()), "g")
    run(6)
    for _ in range(3): run(7)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run(7)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, M, N, K, tr, ns in (("NN 25600x3072x512", 25600, 3072, 512, 0, 1), ("NT 25600x512x3072", 25600, 512, 3072, 1, 1), ("TN 512x3072x25600 5 slabs", 512, 3072, 25600, 2, 5),
                              ("TN 512x3072x25600 16 slabs", 512, 3072, 25600, 2, 16), ("NN 8192x4096x4096", 8192, 4096, 4096, 0, 1)):
    us = t(M, N, K, tr, ns)
    print(f"{name:30s} {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s", flush=True)
