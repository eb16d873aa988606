"""The f16-pair plane GEMM (cfg 6 = scale + split + GEMM, cfg 7 = the GEMM alone on the previous call's pairs) against the
bf16-triple instance (cfg 4 / 5) and the fp32 MFMA instance (cfg 2): error against fp64 on mid-size shapes in every
orientation, then the config-5 shard's shapes timed."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from gemm_bench import bench
from gmvae_amd import _lib as L

def accuracy(M, N, K, trans, ns, scale_a=1.0, scale_b=1.0):
    rng = np.random.default_rng(M + N + K + trans)
    if trans == 2:
        A, W = rng.normal(size=(K, M)) * scale_a, rng.normal(size=(K, N)) * scale_b
    else:
        A, W = rng.normal(size=(M, K)) * scale_a, (rng.normal(size=(K, N)) if trans == 0 else rng.normal(size=(N, K))) * scale_b
    A, W = A.astype(np.float32), W.astype(np.float32)
    A64, W64 = A.astype(np.float64), W.astype(np.float64)
    if trans == 0: ref = A64 @ W64
    elif trans == 1: ref = A64 @ W64.T
    else: ref = np.concatenate([A64.T @ W64, W64.sum(0, keepdims=True)], 0)
    shape = (ns, M + 1, N) if trans == 2 else (M, N)
    Ad, Wd = torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda()
    out = []
    for cfg in (6, 4, 2):
        Cd = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
        L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), 0, L.ptr(Wd), L.ptr(Wd) if trans == 2 else None, L.ptr(Cd), M, N, K, trans, 0, cfg, ns,
                                      L.current_stream()), "gemm_test")
        got = Cd.cpu().numpy().astype(np.float64)
        got = got.sum(axis=0) if trans == 2 else got
        out.append(np.abs(got - ref).max() / np.abs(ref).max())
    return out

for (M, N, K, tr, ns, sa, sb) in [(256, 384, 512, 0, 1, 1, 1), (256, 384, 512, 1, 1, 1, 1), (256, 384, 2048, 2, 4, 1, 1),
                                  (512, 512, 1024, 0, 1, 1e-6, 3e4), (512, 512, 1024, 1, 1, 37.0, 1e-3), (128, 128, 4096, 2, 2, 1e3, 1e-5)]:
    e = accuracy(M, N, K, tr, ns, sa, sb)
    print(f"trans {tr} {M}x{N}x{K} scales {sa:g} {sb:g}: max err / max |ref|  pairs {e[0]:.2e}  triples {e[1]:.2e}  fp32 {e[2]:.2e}", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "acc": sys.exit(0)
shapes = [("fwd_dec  NN 25600x3072x512", 25600, 3072, 512, 0, 1), ("bwd_dX   NT 25600x512x3072", 25600, 512, 3072, 1, 1),
          ("bwd_dW   TN 512x3072x25600", 512, 3072, 25600, 2, 16), ("hidden   NN 25600x512x512", 25600, 512, 512, 0, 1),
          ("square   NN 8192x4096x4096", 8192, 4096, 4096, 0, 1)]
for name, M, N, K, tr, ns in shapes:
    out = []
    for cfg in (4, 5, 6, 7):
        if cfg in (4, 6): bench(M, N, K, tr, False, cfg, ns, iters=2)
        us, tf = bench(M, N, K, tr, False, cfg, ns, iters=20) if cfg in (5, 7) else (0, 0)
        if cfg in (5, 7): out.append(f"cfg{cfg} {us:8.1f} us {tf:6.1f} TF")
    print(f"{name:30s}: " + " | ".join(out), flush=True)
