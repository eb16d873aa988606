"""mega3v_step (the VAE family's step as ONE launch) against mega2v_fwd_bwd + dw_adam (GMVAE_NO_FUSE=1) on the same batches:
parameters and both Adam moments must agree BIT FOR BIT.  argv: [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
G = 16
bad = 0
CASES = [("vae", 2, 1, 100), ("vae_gmp", 64, 10, 256), ("vae", 2, 1, 16), ("vae_gmp", 64, 10, 576), ("vae", 2, 1, 333)]
if len(sys.argv) > 2: CASES = CASES[:int(sys.argv[2])]
for model, Lz, K, B in CASES:
    rng = np.random.default_rng(B)
    xs = torch.from_numpy((rng.random((G, B, 784)) < 0.87).astype(np.uint8)).cuda()
    res = []
    for fused in (True, False):
        if fused: os.environ.pop("GMVAE_NO_FUSE", None)
        else: os.environ["GMVAE_NO_FUSE"] = "1"
        e = Engine(model, 784, Lz, K, [64], random_seed=5)
        sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
        sx.copy_(xs)
        for _ in range(steps // G): replay()
        torch.cuda.synchronize()
        names = [nm for nm, _, _, _ in e.profile_train_levels(xs[0], lr=1e-3, iters=2)]
        res.append((e.params.detach().clone(), e.m.clone(), e.v.clone(), e.handoff_timeouts(), names, e.grads[e.P:e.P + 5].clone()))
    os.environ.pop("GMVAE_NO_FUSE", None)
    same = all(torch.equal(a, b) for a, b in zip(res[0][:3], res[1][:3]))
    d = (res[0][0] - res[1][0]).abs().max().item()
    print(f"{model} L={Lz} B={B:4d} steps={steps // G * G}: {res[0][4]} == {res[1][4]} bit for bit: {same} (max |dtheta| {d:.3e}) timeouts {res[0][3]} / {res[1][3]} "
          f"finite {bool(torch.isfinite(res[0][0]).all())} tails {res[0][5].tolist()[:2]} {res[1][5].tolist()[:2]}", flush=True)
    bad += (not same) or res[0][3] != 0
sys.exit(1 if bad else 0)
