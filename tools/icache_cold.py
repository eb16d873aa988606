"""What a COLD instruction cache costs the one-launch step: the in-kernel span of mega3_step (device clock stamps, gmvae_train_profile)
in a replayed three-step graph, as it is and with ~100 KB of foreign straight-line code run on every CU in front of each stamped
step (GMVAE_ICACHE_FLUSH=1: csrc/kernels.hpp icache_flush).  Beside profiles/round6_pmc_ifetch.txt (3,615 instruction-cache
misses per launch chip-wide out of 1.49 M requests) this bounds what the warm launch's misses can cost.
    python tools/icache_cold.py [batch] [model]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = sys.argv[2] if len(sys.argv) > 2 else "gmvae"
Lz, K = (64, 10) if model != "vae" else (2, 1)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
for rep in range(3):
    for flush in (0, 1):
        if flush:
            os.environ["GMVAE_ICACHE_FLUSH"] = "1"
        else:
            os.environ.pop("GMVAE_ICACHE_FLUSH", None)
        e = Engine(model, 784, Lz, K, [64], random_seed=0)
        e.profile_train_levels(x, iters=10)
        lev = e.profile_train_levels(x, iters=40)
        print(f"flush={flush}  " + "  ".join(f"{nm}: in-kernel span {us:.2f} us, timeline share {tl:.2f} us" for nm, us, _, tl in lev), flush=True)
