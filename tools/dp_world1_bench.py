"""The data-parallel train graph with a ONE-rank RCCL communicator on this GPU: what a rank's step costs besides the
wire time of the all-reduce (the collective is an identity copy here)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
B, G = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 40
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
e.enable_rccl()
sx, replay = e.capture_train_step(B, lr=1e-3, all_reduce=True, n_steps=G)
sx.copy_(torch.from_numpy((np.random.default_rng(0).random((G, B, 784)) < 0.87).astype(np.uint8)).cuda())
print("dp mode:", e.dp_mode)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    replay(); torch.cuda.synchronize()
n = 2000 // G
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): replay()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{dt / (n * G) * 1e6:.2f} us/step  ({B * n * G / dt / 1e6:.2f} M samples/s per rank), loss {e.grads[e.P].item() / B:.3f}")
