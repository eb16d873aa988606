"""A few eager training steps for rocprofv3 (--kernel-trace / --pmc) runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
e = Engine("gmvae", 784, 64, 10, [H], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
for _ in range(n):
    e.train_step(x)
torch.cuda.synchronize()
print("done", e.grads[e.P].item() / B)
