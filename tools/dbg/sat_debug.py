"""Debug: per-tensor gradient errors of the failing saturated cases (prints everything, asserts nothing)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle as O
import hip_util as H
import test_saturated as T
import ctypes as C
from gmvae_amd import _lib as L

def ws_read(ws, cd, model, name, rows, cols):
    off = C.c_uint64()
    L.check(L.lib.gmvae_workspace_offset(C.byref(cd), model, name.encode(), C.byref(off)), name)
    return ws[off.value // 4: off.value // 4 + rows * cols].view(rows, cols).cpu().numpy().astype(np.float64)

def report(model, d, flat, x, eps, u, gs, tail, masks, ws=None, cd=None):
    B = x.shape[0]
    p32 = O.unpack(model, d, flat.astype(np.float64))
    Cc, g = O.loss_and_grads(model, d, p32, x, eps, u, np.float64)
    print("loss", tail[0] / B, Cc["loss"], "nll", tail[1] / B, Cc["nll"], "kl", tail[2] / B, Cc["kl"], "nent", tail[3] / B, Cc["nent"])
    nflip = 0
    for net, ms in masks.items():
        for i in range(1, len(ms)):
            pre, mag = Cc["pre"][net][i - 1]
            diff = ms[i] != (pre > 0)
            if diff.any():
                print("  mask diffs", net, i, int(diff.sum()), "max |pre|/mag", (np.abs(pre[diff]) / mag[diff]).max())
                nflip += diff.sum()
    _, g2 = O.loss_and_grads(model, d, p32, x, eps, u, np.float64, relu_masks=masks)
    for (name, e1), (_, e2) in zip(T._grad_errs(model, d, gs, g, B), T._grad_errs(model, d, gs, g2, B)):
        print(f"  {name:36s} err {e1:.3e}   with device masks {e2:.3e}   max|g| {np.abs(g[name]).max():.3e}")
    print("  max|lam|", np.abs(Cc["lam"]).max(), "sig_q", Cc["sig_q"].min(), Cc["sig_q"].max(), "y max", Cc["y"].max(axis=1).mean() if "y" in Cc else None,
          "|logits| max", np.abs(Cc["logits"]).max() if "logits" in Cc else None)
    return Cc

which = sys.argv[1] if len(sys.argv) > 1 else "eager"
if which == "eager":
    import dataclasses
    name, d, B, sched = T.EAGER[0]
    d = dataclasses.replace(d, **T.FACTORY)
    model = O.MODEL_NAMES[name]
    rng = np.random.default_rng(B + d.L)
    x, eps, u = O.make_inputs(d, B, model)
    p = T.saturate(model, d, O.init_params(model, d, rng), rng, x, span=8.0)
    u[0, 0], u[1, d.K - 1], u[2, :] = O.TINY_F32, T.U_MAX, T.U_MAX
    u[3, :] = O.TINY_F32
    u[B * d.S - 1, 1] = O.TINY_F32
    flat = O.pack(model, d, p, np.float32)
    gs, tail, masks = H.hip_step(model, d, flat, x, eps, u, want_masks=True)
    Cc = report(model, d, flat, x, eps, u, gs, tail, masks)
    # which rows carry the error? bias gradient of encoder_y layer 0 = sum over rows of dhy1
else:
    from gmvae_amd.data import DeviceDataset
    from gmvae_amd.engine import Engine
    from test_timed_path import _device_masks, _noise
    model, Lz, K, H_, B = "gmvae", 128, 10, 512, 64
    rng = np.random.default_rng(7)
    pix, _ = T._structured_pixels(8192, 784, rng)
    ds = DeviceDataset(pix, shuffle=True, seed=3)
    e = Engine(model, 784, Lz, K, [H_], random_seed=5)
    replay = e.capture_train_pipeline(ds, B, lr=3e-3, n_steps=16)
    for _ in range(19):
        replay()
    torch.cuda.synchronize()
    flat = e.params.detach().cpu().numpy().copy()
    xb = (pix[:B].astype(np.float32) / 255.0 < rng.random((B, 784), dtype=np.float32)).astype(np.uint8)
    mid = O.MODEL_NAMES[model]
    d = O.Dims(D=784, L=Lz, K=K, hidden=(H_,))
    for mode in ("graph", "eager"):
        e2 = Engine(model, 784, Lz, K, [H_], random_seed=5)
        with torch.no_grad():
            e2.params.copy_(torch.from_numpy(flat).cuda())
        e2.global_step = 304; e2.step_dev.fill_(304)
        eps, u = _noise(L, B, Lz, K, 0, e2.noise_seed, 304, True)
        if mode == "graph":
            sx, rp = e2.capture_train_step(B, lr=1e-3, n_steps=1)
            sx.copy_(torch.from_numpy(xb).cuda()); rp(); torch.cuda.synchronize()
        else:
            e2.step(torch.from_numpy(xb).cuda(), torch.from_numpy(eps).cuda(), torch.from_numpy(u).cuda()); torch.cuda.synchronize()
        got = e2.grads.cpu().numpy().astype(np.float64)
        masks = _device_masks(e2, mid, d, B)
        print("====", mode, L.step_schedule(e2.dims(B), mid))
        Cc = report(mid, d, flat, xb, eps, u, got[:e2.P], got[e2.P:], masks)
        cd = e2.dims(B); ws = e2._ws[(B, 1)]
        for nm, ref, rows, cols in (("y", Cc["y"], B, 12), ("logits", Cc["logits"], B, 12)):
            try:
                a = ws_read(ws, cd, mid, nm, rows, cols)[:, :K]
                print("  ws", nm, "max abs err", np.abs(a - ref).max(), "max ref", np.abs(ref).max())
            except Exception as ex:
                print("  ws", nm, ex)
