"""Debug aid: one step of a 1-step train graph (mega2v / mega2 path) against the fp64 oracle, intermediate by intermediate.
argv: model L K B [seed]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle as O
import test_timed_path as T
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
model, Lz, K, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 11
mid = O.MODEL_NAMES[model]
d = O.Dims(D=784, L=Lz, K=K, hidden=(64,))
e = Engine(model, 784, Lz, K, [64], random_seed=seed)
print("schedule", L.step_schedule(e.dims(B), mid))
flat0 = e.params.detach().cpu().numpy().astype(np.float64)
x = (np.random.default_rng(B).random((B, 784)) < 0.87).astype(np.uint8)
sx, replay = e.capture_train_step(B, lr=1e-3, n_steps=1)
sx.copy_(torch.from_numpy(x).cuda()); replay(); torch.cuda.synchronize()
print("timeouts", e.handoff_timeouts())
eps, u = T._noise(L, B, Lz, K, 0, e.noise_seed, 0, mid == O.MODEL_GMVAE)
Cc, g = O.loss_and_grads(mid, d, O.unpack(mid, d, flat0), x, eps, u, np.float64)
ws = e._ws[(B, 1)]
def buf(name, rows, cols):
    off = C.c_uint64()
    L.check(L.lib.gmvae_workspace_offset(C.byref(e.dims(B)), mid, name.encode(), C.byref(off)), name)
    return ws[off.value // 4: off.value // 4 + rows * cols].view(rows, cols).cpu().numpy().astype(np.float64)
def cmp(name, got, ref):
    err = np.abs(got - ref)
    i = np.unravel_index(err.argmax(), err.shape)
    print(f"  {name:8s} max err {err.max():.3e} (ref max {np.abs(ref).max():.3e}) at {i}; rows with err > 1e-4 max: {np.unique(np.where(err > 1e-4 * max(np.abs(ref).max(), 1e-9))[0])[:20]}")
enc = "encoder_y" if mid == O.MODEL_GMVAE else "encoder"
hs = Cc["hs_y"] if mid == O.MODEL_GMVAE else Cc["hs_e"]
cmp("hy1", buf("hy1", B, 64), hs[1])
cmp("z", buf("z", B, Lz), Cc["z"])
cmp("hd1", buf("hd1", B, 64), Cc["hs_d"][1])
cmp("g", buf("g", B, 784), Cc["dlam"] * B)
cmp("dqp", buf("dqp", B, 2 * Lz), Cc["dqp"] * B)
cmp("logq", buf("logq", 1, B), Cc["logq"][None])
cmp("logp", buf("logp", 1, B), Cc["logp"][None])
cmp("logpx", buf("logpx", 1, B), Cc["logpx"][None])
P = e.P
gb = e.grads.cpu().numpy().astype(np.float64)
lay, _, _ = O.param_layout(mid, d)
for name, shape, off in lay:
    k = int(np.prod(shape)); ref = g[name].ravel()
    print(f"  grad {name:30s} rel-to-max err {np.abs(gb[off:off + k] / B - ref).max() / max(np.abs(ref).max(), 1e-9):.2e}")
masks = T._device_masks(e, mid, d, B)
for net, ms in masks.items():
    for i in range(1, len(ms)):
        pre, mag = Cc["pre"][net][i - 1]
        diff = ms[i] != (pre > 0)
        print(f"  mask {net} layer {i}: {diff.sum()} differing units of {diff.size}; shapes {ms[i].shape} {pre.shape}",
              (np.abs(pre[diff]) / mag[diff])[:5] if diff.any() else "")
        if diff.any():
            r, c = np.where(diff)
            print("    first:", r[:5], c[:5], "device h", buf("hy1" if net != "decoder" else "hd1", B, 64)[r[:5], c[:5]], "pre64", pre[r[:5], c[:5]])
