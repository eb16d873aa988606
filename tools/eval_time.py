"""Forward-only pass times (Engine.profile_forward: captured passes replayed back to back) with csrc/evalf.hpp and, for comparison,
with GMVAE_NO_EVALF=1 (the chain / general schedules):   python tools/eval_time.py [model] [latent] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
model = sys.argv[1] if len(sys.argv) > 1 else "gmvae"
Lz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
for B, S in ((1024, 1), (256, 1), (64, 1), (8192, 1), (1024, 5), (1024, 50), (100, 50), (4096, 10)):
    x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
    out = []
    for no in ("", "1"):
        if no:
            os.environ["GMVAE_NO_EVALF"] = "1"
        else:
            os.environ.pop("GMVAE_NO_EVALF", None)
        e = Engine(model, 784, Lz, K, [64], n_samples=S, random_seed=0)
        e.profile_forward(x, n_samples=S, iters=50)
        lev, us, _ = e.profile_forward(x, n_samples=S, iters=400)
        out.append((us, len(lev)))
    print(f"{model} L={Lz} B={B:5d} S={S:3d}: evalf {out[0][0]:8.2f} us ({out[0][1]} launches)   without {out[1][0]:8.2f} us ({out[1][1]} launches)", flush=True)
