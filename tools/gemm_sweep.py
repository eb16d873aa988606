import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
rng = np.random.default_rng(0)
bad = 0
for trans in (2, 0, 1):
    for M, N, K in itertools.product((5, 16, 33, 64, 65), (3, 16, 20, 32, 48), (4, 8, 30, 64, 130)):
        for cfg in (0, 1, 2):
            ns = 2 if trans == 2 else 1
            if trans == 2:
                A = rng.normal(size=(K, M)).astype(np.float32); W = rng.normal(size=(K, N)).astype(np.float32)
                ref = np.concatenate([A.astype(np.float64).T @ W, W.astype(np.float64).sum(0, keepdims=True)], 0)
                Cd = torch.full((ns, M + 1, N), float("nan"), dtype=torch.float32, device="cuda")
            elif trans == 0:
                A = rng.normal(size=(M, K)).astype(np.float32); W = rng.normal(size=(K, N)).astype(np.float32)
                ref = A.astype(np.float64) @ W; Cd = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
            else:
                A = rng.normal(size=(M, K)).astype(np.float32); W = rng.normal(size=(N, K)).astype(np.float32)
                ref = A.astype(np.float64) @ W.astype(np.float64).T; Cd = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
            Ad, Wd = torch.from_numpy(A).cuda(), torch.from_numpy(W).cuda()
            L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), 0, L.ptr(Wd), L.ptr(Wd) if trans == 2 else None, L.ptr(Cd), M, N, K, trans, 0, cfg, ns, L.current_stream()), "g")
            got = Cd.cpu().numpy().astype(np.float64)
            if trans == 2: got = got.sum(0)
            err = np.abs(got - ref).max() if np.isfinite(got).all() else float("inf")
            if not err < 1e-3:
                bad += 1
                if bad < 25: print("BAD trans", trans, "M,N,K", M, N, K, "cfg", cfg, "err", err)
print("bad", bad)
