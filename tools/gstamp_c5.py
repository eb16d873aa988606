import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gmvae_amd.engine import Engine
e = Engine("gmvae", 3072, 64, 64, [512], n_samples=50, random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((512, 3072)) < 0.87).astype(np.uint8)).cuda()
for _ in range(3):
    e.train_step(x)
torch.cuda.synchronize()
