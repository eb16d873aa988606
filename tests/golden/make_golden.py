"""Regenerates tests/golden/*.npz from the fp64 oracle (run from the repo root:
``python tests/golden/make_golden.py``).

The reference cannot run here, so these vectors are produced
by the oracle, not by TF: they pin the oracle against regressions and give the
GPU parity tests fixed (params, x, eps, u) -> (nll, kl, nent, loss, grads)
cases that do not depend on RNG implementation details.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402

CASES = {
    # name: (model, Dims, B)
    "gmvae_d784_k10_l8_h64_b8": ("gmvae", O.Dims(D=784, L=8, K=10, hidden=(64,)), 8),
    "gmvae_d200_k10_l64_h64_b6_ragged": ("gmvae", O.Dims(D=200, L=64, K=10, hidden=(64,)), 6),
    "gmvae_d100_k7_l5_h24x2_b8": ("gmvae", O.Dims(D=100, L=5, K=7, hidden=(24, 24)), 8),
    "gmvae_iwae_d96_k6_l4_h16_s3_b4": ("gmvae", O.Dims(D=96, L=4, K=6, hidden=(16,), S=3), 4),
    "vae_d784_l2_h64_b8": ("vae", O.Dims(D=784, L=2, K=1, hidden=(64,)), 8),
    "vae_gmp_d784_k10_l64_h64_b8": ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(64,)), 8),
}


def build(name):
    mname, d, B = CASES[name]
    model = O.MODEL_NAMES[mname]
    rng = np.random.default_rng(sum(map(ord, name)))
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    p = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}   # stored as fp32, exactly
    x, eps, u = O.make_inputs(d, B, model, seed_x=11, seed_noise=13)
    C, g = O.loss_and_grads(model, d, p, x, eps, u, np.float64)
    out = dict(model=np.int32(model), D=d.D, L=d.L, K=d.K, S=d.S, hidden=np.array(d.hidden, np.int32),
               x=x, eps=eps, params=O.pack(model, d, p, np.float32), grads=O.pack(model, d, g, np.float32),
               nll=C["nll"], kl=C["kl"], nent=C["nent"], loss=C["loss"], logw=C["logw"])
    if u is not None:
        out["u"] = u
    return out


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **build(name))
        print("wrote", name)
