"""Diagnostic for a trajectory case (tests/test_timed_path.py): per seed, the last-step gradient error of the device (a) against
the fp64 oracle TRAJECTORY (what the test gates) and (b) against the fp64 oracle evaluated at the DEVICE's own parameters before
the last step -- (b) is the kernels' error alone, (a) adds what n - 1 steps of Adam drift do to the gradient.
argv: model D L K H B n seed [seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle as O
import test_timed_path as T
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
model, D, Lz, K, H, B, n = sys.argv[1], *map(int, sys.argv[2:8])
for seed in map(int, sys.argv[8:]):
    mid = O.MODEL_NAMES[model]
    d = O.Dims(D=D, L=Lz, K=K, hidden=(H,))
    xs = (np.random.default_rng(B).random((n, B, D)) < 0.87).astype(np.uint8)
    xd = torch.from_numpy(xs).cuda()
    e1 = Engine(model, D, Lz, K, [H], random_seed=seed)
    flat0 = e1.params.detach().cpu().numpy()
    sx1, replay1 = e1.capture_train_step(B, lr=T.LR, n_steps=1)
    masks, pres = [], []
    for t in range(n):
        pres.append(e1.params.detach().cpu().numpy().astype(np.float64))
        sx1.copy_(xd[t]); replay1(); torch.cuda.synchronize()
        masks.append(T._device_masks(e1, mid, d, B))
    flat_ref, Cc, g, gs = T._oracle_trajectory(L, mid, d, flat0, xs, e1.noise_seed, masks_of_step=lambda t: masks[t], params_of_step=lambda t: pres[t], tag=f"seed{seed}")
    pre = pres[n - 1]
    eps, u = T._noise(L, B, Lz, K, 0, e1.noise_seed, n - 1, mid == O.MODEL_GMVAE)
    C2, g2 = O.loss_and_grads(mid, d, O.unpack(mid, d, pre), xs[n - 1], eps, u, np.float64, relu_masks=masks[n - 1])
    g2 = O.pack(mid, d, g2, np.float64)
    buf = e1.grads.cpu().numpy().astype(np.float64)
    lay, P, _ = O.param_layout(mid, d)
    print(f"seed {seed}: loss dev {buf[P] / B:.6f} traj {Cc['loss']:.6f} at-device-params {C2['loss']:.6f}")
    for name, shape, off in lay:
        k = int(np.prod(shape))
        got = buf[off:off + k] / B
        ea = np.abs(got - g[off:off + k]).max() / max(np.abs(g[off:off + k]).max(), 1e-6)
        eb = np.abs(got - g2[off:off + k]).max() / max(np.abs(g2[off:off + k]).max(), 1e-6)
        dp = np.abs(pre[off:off + k] - 0).max()
        print(f"   {name:32s} vs trajectory {ea:.2e}   vs oracle at device params {eb:.2e}")
