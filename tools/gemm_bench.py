"""Micro-benchmark of the grouped GEMM kernel through gmvae_gemm_test (torch events, one process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L

def bench(M, N, K, trans, u8, cfg, ns=1, iters=200):
    rng = np.random.default_rng(0)
    if trans == 2:
        A = torch.from_numpy((rng.random((K, M)) < 0.5).astype(np.uint8)).cuda() if u8 else torch.randn(K, M, device="cuda")
        W = torch.randn(K, N, device="cuda")
        C = torch.empty(ns, M + 1, N, device="cuda")
    else:
        A = torch.from_numpy((rng.random((M, K)) < 0.5).astype(np.uint8)).cuda() if u8 else torch.randn(M, K, device="cuda")
        W = torch.randn(K, N, device="cuda") if trans == 0 else torch.randn(N, K, device="cuda")
        C = torch.empty(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    def run():
        L.check(L.lib.gmvae_gemm_test(L.ptr(A), int(u8), L.ptr(W), L.ptr(b), L.ptr(C), M, N, K, trans, 0, cfg, ns, L.current_stream()), "g")
    for _ in range(10): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return us, 2.0 * M * N * K / us * 1e-6

if __name__ == "__main__":
    shapes = [("fwd_x  NN u8 1024x64x784", 1024, 64, 784, 0, True), ("fwd_dec NN 1024x784x64", 1024, 784, 64, 0, False),
              ("dX NT 1024x64x784", 1024, 64, 784, 1, False), ("dW TN 64x784x1024 ns4", 64, 784, 1024, 2, False),
              ("dW TN u8 784x64x1024 ns4", 784, 64, 1024, 2, True), ("NN 4096x512x512", 4096, 512, 512, 0, False),
              ("NN 8192x4096x4096", 8192, 4096, 4096, 0, False), ("tiny NN 1024x10x64", 1024, 10, 64, 0, False)]
    for name, M, N, K, tr, u8 in shapes:
        for cfg in (0, 1, 2):
            if M * N * K > 1e10 and cfg == 0: continue
            ns = 4 if tr == 2 else 1
            us, tf = bench(M, N, K, tr, u8, cfg, ns, iters=50 if M * N * K > 1e9 else 200)
            print(f"{name:28s} cfg{cfg}: {us:9.2f} us  {tf:8.2f} TFLOP/s", flush=True)
