"""Helpers for the -m gpu parity tests: call the C ABI with torch device buffers."""
import ctypes as C

import numpy as np
import torch

import oracle as O
from gmvae_amd import _lib as L


def dims_of(d: O.Dims, B: int):
    gb = np.asarray(d.gen_bias_init, np.float32)
    vec = torch.from_numpy(gb.copy()).cuda() if gb.ndim else None        # vector bias_init (ABI v3)
    cd = L.make_dims(B, d.D, d.L, d.K, d.hidden, S=d.S, sigma_min=d.sigma_min, raw_sigma_bias=d.raw_sigma_bias,
                     temperature=d.temperature, gen_bias_init=0.0 if gb.ndim else float(gb), gen_bias_vec=vec)
    cd._keep_vec = vec                                                    # the struct holds a raw device pointer
    return cd


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def workspace(cd, model):
    return torch.zeros(L.workspace_bytes(cd, model) // 4 + 64, dtype=torch.float32, device="cuda")


def hip_step(model, d: O.Dims, flat, x, eps, u, seed=0, step=0):
    """Returns (grads_sum[P_pad] float64 numpy, tail[8])."""
    B = x.shape[0]
    cd = dims_of(d, B)
    P, _ = L.param_count(cd, model)
    params = dev(flat, torch.float32)
    xd = dev(x, torch.uint8)
    ed = None if eps is None else dev(eps, torch.float32)
    ud = None if u is None else dev(u, torch.float32)
    grads = torch.full((P + L.TAIL,), float("nan"), dtype=torch.float32, device="cuda")
    ws = workspace(cd, model)
    rc = L.lib.gmvae_step(C.byref(cd), model, L.ptr(xd), L.ptr(ed), L.ptr(ud), L.ptr(params), L.ptr(grads), L.ptr(ws),
                          seed, step, None, L.current_stream())
    L.check(rc, "gmvae_step")
    torch.cuda.synchronize()
    g = grads.cpu().numpy().astype(np.float64)
    return g[:P], g[P:]


def hip_forward(model, d: O.Dims, flat, x, eps, u, want_rows=True):
    B = x.shape[0]
    cd = dims_of(d, B)
    R = B * d.S
    params, xd = dev(flat, torch.float32), dev(x, torch.uint8)
    ed = None if eps is None else dev(eps, torch.float32)
    ud = None if u is None else dev(u, torch.float32)
    tail = torch.zeros(L.TAIL, dtype=torch.float32, device="cuda")
    rows = torch.zeros(R, 4, dtype=torch.float32, device="cuda")
    z = torch.zeros(R, d.L, dtype=torch.float32, device="cuda")
    y = torch.zeros(R, d.K, dtype=torch.float32, device="cuda")
    lg = torch.zeros(B, d.K, dtype=torch.float32, device="cuda")
    ws = workspace(cd, model)
    rc = L.lib.gmvae_forward(C.byref(cd), model, L.ptr(xd), L.ptr(ed), L.ptr(ud), L.ptr(params), L.ptr(tail),
                             L.ptr(rows), L.ptr(z), L.ptr(y), L.ptr(lg), L.ptr(ws), 0, 0, L.current_stream())
    L.check(rc, "gmvae_forward")
    torch.cuda.synchronize()
    return tail.cpu().numpy(), rows.cpu().numpy(), z.cpu().numpy(), y.cpu().numpy(), lg.cpu().numpy()


MARGINS = []      # (what, worst error / its gate) of every compare_step call: tests/test_hip_parity.py prints the maxima


def compare_step(model, d, p, x, eps, u, loss_rtol=1e-4, grad_rtol=1e-4):
    """HIP step vs the fp64 oracle on identical (params, x, eps, u).  Gates: the ELBO at loss_rtol
    relative (north_star's 1e-4), and EACH term relative to ITSELF -- |d nll| <= 1e-4 |nll|, |d kl| <= 1e-4 max(|kl|, 1),
    |d nent| <= 1e-4 max(|nent|, 1) -- so that the O(1-10) kl and entropy terms cannot hide inside the budget of an
    O(500) loss; every gradient tensor at grad_rtol of its own max."""
    B = x.shape[0]
    flat = O.pack(model, d, p, np.float32)
    p32 = O.unpack(model, d, flat.astype(np.float64))            # the values the GPU actually sees
    Cc, g = O.loss_and_grads(model, d, p32, x, eps, u, np.float64)
    gs, tail = hip_step(model, d, flat, x, eps, u)
    assert tail[4] == B
    loss = tail[0] / B
    assert abs(loss - Cc["loss"]) <= loss_rtol * abs(Cc["loss"]), (loss, Cc["loss"])
    terms = {"nll": (tail[1] / B, Cc["nll"], loss_rtol * abs(Cc["nll"])),
             "kl": (tail[2] / B, Cc["kl"], 1e-4 * max(abs(Cc["kl"]), 1.0)),
             "nent": (tail[3] / B, Cc["nent"], 1e-4 * max(abs(Cc["nent"]), 1.0))}
    MARGINS.append(("loss", abs(loss - Cc["loss"]) / (loss_rtol * abs(Cc["loss"]))))
    for nm, (got, ref, gate) in terms.items():
        MARGINS.append((nm, abs(got - ref) / gate))
        assert abs(got - ref) <= gate, f"{nm}: {got} vs {ref} (gate {gate:.2e})"
    lay, P, _ = O.param_layout(model, d)
    worst = 0.0
    for name, shape, off in lay:
        n = int(np.prod(shape))
        got = gs[off:off + n].reshape(shape) / B
        ref = g[name]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(got - ref).max() / scale
        worst = max(worst, err)
        MARGINS.append(("grad S>1" if d.S > 1 else "grad", err / grad_rtol))
        assert err <= grad_rtol, f"{name}: rel-to-max err {err:.3e}"
    return loss, worst
