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
                     temperature=d.temperature, gen_bias_init=0.0 if gb.ndim else float(gb), gen_bias_vec=vec,
                     hidden_act=getattr(d, "act", "relu"))
    cd._keep_vec = vec                                                    # the struct holds a raw device pointer
    return cd


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def workspace(cd, model):
    return torch.zeros(L.workspace_bytes(cd, model) // 4 + 64, dtype=torch.float32, device="cuda")


NETS = {O.MODEL_GMVAE: (("encoder_y", "he", False), ("encoder_gmm", "hg", True), ("decoder", "hd", True)),
        O.MODEL_VAE: (("encoder", "he", False), ("decoder", "hd", True)),
        O.MODEL_VAE_GMP: (("encoder", "he", False), ("decoder", "hd", True))}
PRE_TOL = 1e-5         # a ReLU may take the other side than in fp64 only where |pre-activation| <= PRE_TOL * sum_k |a_k| |w_kj|
FLIPS = []             # (what, net, |pre| / sum |a||w|) of every unit where a compared step's ReLU mask differed from fp64's


def device_masks(ws, cd, model, d, B):
    """The ReLU masks of the step that last ran in workspace `ws` (a float32 device tensor): (kept activation > 0) of every
    hidden layer (gmvae_workspace_offset "he<i>" / "hg<i>" / "hd<i>"; rows = B for the encoder of x, B*S otherwise)."""
    out = {}
    for net, tag, per_sample in NETS[model]:
        ms = [None]
        for i, h in enumerate(d.hidden, start=1):
            off = C.c_uint64()
            L.check(L.lib.gmvae_workspace_offset(C.byref(cd), model, f"{tag}{i}".encode(), C.byref(off)), f"offset {tag}{i}")
            rows = B * d.S if per_sample else B
            ms.append(ws[off.value // 4: off.value // 4 + rows * h].view(rows, h).cpu().numpy() > 0)
        out[net] = ms
    return out


def check_masks(masks, pres, what):
    """Every unit where the device's mask differs from the fp64 one must be numerically zero in fp64 (PRE_TOL); returns the
    number of such units."""
    n = 0
    for net, ms in masks.items():
        for i in range(1, len(ms)):
            pre, mag = pres[net][i - 1]
            diff = ms[i] != (pre > 0)
            if diff.any():
                ratio = np.abs(pre[diff]) / np.maximum(mag[diff], 1e-30)
                FLIPS.extend((what, net, float(r)) for r in ratio)
                n += int(diff.sum())
                assert ratio.max() <= PRE_TOL, (f"{what} {net} layer {i}: the device's ReLU mask differs from fp64's at a "
                                                f"pre-activation that is NOT numerically zero (|pre| / sum|a||w| = {ratio.max():.2e})")
    return n


def hip_step(model, d: O.Dims, flat, x, eps, u, seed=0, step=0, want_masks=False):
    """Returns (grads_sum[P_pad] float64 numpy, tail[8]) (+ the step's ReLU masks with want_masks)."""
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
    if want_masks:
        return g[:P], g[P:], device_masks(ws, cd, model, d, B)
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
    O(500) loss; every gradient tensor at grad_rtol of its own max.

    ReLU has no derivative at 0, and an fp32 pre-activation that is zero to within rounding can land on the other side than
    the fp64 one (one unit in ~10^5 at these sizes): the oracle takes the device's subgradient there -- and only there:
    check_masks asserts that every unit whose mask differs is numerically zero in fp64 -- so that any seed runs inside the
    same gates (tests/test_timed_path.py does the same over trajectories)."""
    B = x.shape[0]
    flat = O.pack(model, d, p, np.float32)
    p32 = O.unpack(model, d, flat.astype(np.float64))            # the values the GPU actually sees
    Cc, g = O.loss_and_grads(model, d, p32, x, eps, u, np.float64)
    gs, tail, masks = hip_step(model, d, flat, x, eps, u, want_masks=True)
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

    def grad_errs(g):
        out = []
        for name, shape, off in lay:
            n = int(np.prod(shape))
            got, ref = gs[off:off + n].reshape(shape) / B, g[name]
            out.append((name, np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)))
        return out

    errs = grad_errs(g)
    if max(e for _, e in errs) > grad_rtol and getattr(d, "act", "relu") == "relu":
        # a gradient outside its gate: the device's ReLU took another side than fp64 somewhere?  Legitimate only at units that are
        # numerically zero in fp64 (check_masks asserts it); the oracle then takes the device's subgradients and the gates apply
        who = [k for k, v in O.MODEL_NAMES.items() if v == model][0]
        if check_masks(masks, Cc["pre"], f"{who} B={B} D={d.D} L={d.L} K={d.K} H={d.hidden} S={d.S}"):
            _, g = O.loss_and_grads(model, d, p32, x, eps, u, np.float64, relu_masks=masks)
            errs = grad_errs(g)
    worst = 0.0
    for name, err in errs:
        worst = max(worst, err)
        MARGINS.append(("grad S>1" if d.S > 1 else "grad", err / grad_rtol))
        assert err <= grad_rtol, f"{name}: rel-to-max err {err:.3e}"
    return loss, worst
