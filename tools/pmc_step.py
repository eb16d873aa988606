"""A few training steps for rocprofv3 (--kernel-trace / --pmc) runs: argv = hidden, batch, n, [steps per graph launch], [latent],
[data size], [components], [samples].
With a 4th argument > 0 the steps run as multi-step train graphs (the bench's schedule: first layer inside
mega_fwd_bwd from the second step of a launch on); otherwise eager train_step calls."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
G = int(sys.argv[4]) if len(sys.argv) > 4 else 0
Lz = int(sys.argv[5]) if len(sys.argv) > 5 else 64
D = int(sys.argv[6]) if len(sys.argv) > 6 else 784
K = int(sys.argv[7]) if len(sys.argv) > 7 else 10
S = int(sys.argv[8]) if len(sys.argv) > 8 else 1
e = Engine("gmvae", D, Lz, K, [H], n_samples=S, random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, D)) < 0.87).astype(np.uint8)).cuda()
if G > 0:
    sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
    sx.copy_(x.unsqueeze(0).expand(G, -1, -1))
    for _ in range(n):
        replay()
else:
    for _ in range(n):
        e.train_step(x)
torch.cuda.synchronize()
print("done", e.grads[e.P].item() / B)
