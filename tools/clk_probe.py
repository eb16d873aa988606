import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
e = Engine("gmvae", 784, 128, 10, [512], random_seed=0)
B = 64
sx, replay = e.capture_train_step(B, 1e-3, n_steps=40)
sx.copy_(torch.from_numpy((np.random.default_rng(0).random((40, B, 784)) < 0.87).astype(np.uint8)).cuda())
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5: replay()
torch.cuda.synchronize()
t = e.grads[e.P:].cpu().numpy()
print("shader cycles", t[6], "realtime ticks(100MHz)", t[7], "=> clock GHz", t[6] / max(t[7], 1) * 0.1)
