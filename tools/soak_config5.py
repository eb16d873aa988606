"""The general schedule with the plane GEMMs at the config-5 dims (D = 3072, K = 64, hidden 512, S = 50): a few hundred TF-Adam steps on a
fixed set of batches, twice from the same start -- the two loss curves must be bit-identical (nothing in the schedule depends on
timing) and finite, and the loss must fall.  argv: [steps] [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(0)
proto = rng.random((8, 3072)) < 0.25                       # eight prototypes + pixel noise: something to learn
xs = torch.from_numpy((proto[rng.integers(0, 8, (4, B))] ^ (rng.random((4, B, 3072)) < 0.05)).astype(np.uint8)).cuda()
curves = []
for run in range(2):
    e = Engine("gmvae", 3072, 64, 64, [512], n_samples=50, random_seed=5)
    ls = []
    for t in range(steps):
        tail = e.train_step(xs[t % 4], lr=1e-3)
        if t % 10 == 9 or t == steps - 1:
            tl = tail.cpu().numpy()
            ls.append(float(tl[0] / tl[4]))
    torch.cuda.synchronize()
    curves.append(np.array(ls))
    print(f"run {run}: loss at step 10 {ls[0]:.4f}, at step {steps} {ls[-1]:.4f}", flush=True)
same = np.array_equal(curves[0], curves[1])
ok = same and np.isfinite(curves[0]).all() and curves[0][-1] < curves[0][0]
print("bit-identical curves:", same, " finite:", bool(np.isfinite(curves[0]).all()), " loss fell:", bool(curves[0][-1] < curves[0][0]))
sys.exit(0 if ok else 1)
