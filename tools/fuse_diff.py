"""Per-tensor difference of gradients / parameters between mega3_step and the two-launch form after n steps (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(B)
xs = torch.from_numpy((rng.random((n, B, 784)) < 0.87).astype(np.uint8)).cuda()
out = []
for fused in (True, False):
    if fused: os.environ.pop("GMVAE_NO_FUSE", None)
    else: os.environ["GMVAE_NO_FUSE"] = "1"
    e = Engine("gmvae", 784, 64, 10, [64], random_seed=5)
    sx, replay = e.capture_train_step(B, 1e-3, n_steps=n)
    sx.copy_(xs if n > 1 else xs[0])
    replay(); torch.cuda.synchronize()
    out.append((e._slot_views(e.grads[:e.P]), e.views(), e.grads[e.P:e.P + 8].clone(), e))
for name in out[0][0]:
    g0, g1 = out[0][0][name], out[1][0][name]
    p0, p1 = out[0][1][name], out[1][1][name]
    dg = (g0 - g1).abs().max().item(); dp = (p0 - p1).abs().max().item()
    if dg or dp:
        bad = (g0 != g1).nonzero()
        print(f"{name:40s} grad max|d| {dg:.3e} of max {g1.abs().max().item():.3e}; param max|d| {dp:.3e}; {len(bad)} of {g0.numel()} differ; first {bad[:3].tolist()}")
print("tail fused", out[0][2].tolist(), "\ntail two  ", out[1][2].tolist())
