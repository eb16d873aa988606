"""A few hundred TF-Adam steps of the general schedule with the plane GEMMs forced on (GMVAE_PLANES_MINROWS), once as f16 pairs and
once as exact bf16 triples (GMVAE_PLANES_EXACT=1), same data and seeds: the two loss curves must stay together (the per-product
3 x 2^-22 of the pairs is noise at Adam's scale) and finite.  argv: [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["GMVAE_PLANES_MINROWS"] = "128"
os.environ["GMVAE_NO_SKINNY"] = "1"
from gmvae_amd.engine import Engine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
B, D, S = 64, 384, 4
xs = torch.from_numpy((rng.random((8, B, D)) < 0.3).astype(np.uint8)).cuda()
curves = []
for exact in ("0", "1"):
    os.environ["GMVAE_PLANES_EXACT"] = exact
    e = Engine("gmvae", D, 32, 10, [128], n_samples=S, random_seed=3)
    ls = []
    for t in range(steps):
        tail = e.train_step(xs[t % 8], lr=1e-3)
        if t % 10 == 9 or t == steps - 1:
            tl = tail.cpu().numpy()
            ls.append(float(tl[0] / tl[4]))
    torch.cuda.synchronize()
    curves.append(np.array(ls))
    print("exact=" + exact, "loss at step 10 / last", ls[0], ls[-1], flush=True)
d = np.abs(curves[0] - curves[1]) / np.abs(curves[1])
print("max relative difference between the curves:", d.max(), "finite:", bool(np.isfinite(curves[0]).all() and np.isfinite(curves[1]).all()))
sys.exit(0 if d.max() < 5e-3 and np.isfinite(curves[0]).all() else 1)
