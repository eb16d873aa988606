"""The committed golden vectors still agree with the oracle (fp64 and fp32)."""
import glob
import os

import numpy as np
import pytest

import oracle as O

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load_case(path):
    z = np.load(path)
    d = O.Dims(D=int(z["D"]), L=int(z["L"]), K=int(z["K"]), S=int(z["S"]), hidden=tuple(int(h) for h in z["hidden"]))
    return z, int(z["model"]), d


def test_fixtures_present():
    assert len(FILES) >= 6


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-7), (np.float32, 1e-4)])
def test_oracle_reproduces_golden(path, dtype, tol):
    z, model, d = load_case(path)
    p = O.unpack(model, d, z["params"])
    u = z["u"] if "u" in z.files else None
    C, g = O.loss_and_grads(model, d, p, z["x"], z["eps"], u, dtype)
    ltol = 1e-12 if dtype == np.float64 else tol          # scalars are stored in fp64, grads in fp32
    assert abs(C["loss"] - z["loss"]) <= ltol * abs(z["loss"])
    for k in ("nll", "kl", "nent"):
        assert abs(C[k] - z[k]) <= tol * max(abs(float(z[k])), 1.0)
    gf = O.pack(model, d, g, np.float64)
    scale = max(np.abs(z["grads"]).max(), 1e-3)
    assert np.abs(gf - z["grads"]).max() <= tol * scale * (1 if dtype == np.float64 else 10)
