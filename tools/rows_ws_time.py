"""The config-5 shard's row-panel layers: rows_ws (cfg 8 of gmvae_gemm_test) against the grouped GEMM's best tile configuration, us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gmvae_amd import _lib as L

def t(M, N, K, trans, cfg, side=False, iters=50):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(K, N, device="cuda") if trans == 0 else torch.randn(N, K, device="cuda")
    C = torch.empty(M, N, device="cuda")
    b = torch.randn(M, N, device="cuda") if side else (torch.randn(N, device="cuda") if trans == 0 else None)
    def run():
        L.check(L.lib.gmvae_gemm_test(L.ptr(A), 0, L.ptr(W), L.ptr(b) if b is not None else None, L.ptr(C), M, N, K, trans, 1 if trans == 0 else 0, cfg, 1, L.current_stream()), "g")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

R = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
for name, N, K, tr, side in (("NN y/z -> 512 (K 64)", 512, 64, 0, False), ("NT dhg (K 128, mask)", 512, 128, 1, True), ("NT dy prior (K 128)", 64, 128, 1, False),
                             ("NT dz (K 512)", 64, 512, 1, False), ("NT dy += (K 512, addend)", 64, 512, 1, True)):
    row = [f"rows_ws {t(R, N, K, tr, 8, side):7.1f}"]
    for cfg in (1, 2):
        row.append(f"cfg{cfg} {t(R, N, K, tr, cfg):7.1f}")
    mb = (R * K + R * N * (2 if side else 1)) * 4e-6
    print(f"{name:28s} {mb:6.1f} MB : " + "  ".join(row) + " us", flush=True)
