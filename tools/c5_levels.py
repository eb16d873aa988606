"""The config-5 shard's step launch by launch (gmvae_step_profile: hipEvents around eager launches, mean microseconds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = Engine("gmvae", 3072, 64, 64, [512], n_samples=50, random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 3072)) < 0.3).astype(np.uint8)).cuda()
eng.profile_levels(x, iters=3)
tot = 0.0
for n, us, fl in eng.profile_levels(x, iters=20):
    tot += us
    print(f"{n:34s} {us:8.1f} us" + (f"  {fl / us * 1e-6:7.1f} TFLOP/s" if fl else ""))
print(f"{'sum':34s} {tot:8.1f} us")
