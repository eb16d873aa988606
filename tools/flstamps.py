"""Shader-clock stamps inside mega2_fwd_bwd's first-layer stage (diagnostic path GMVAE_STAMPS=6), medians over workgroups."""
import sys, os, ctypes as C
os.environ["GMVAE_STAMPS"] = "6"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
B = 1024
nP = B // 16
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
G = 16
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
sx.copy_(x.unsqueeze(0).expand(G, -1, -1))
for _ in range(200): replay()
torch.cuda.synchronize()
d, ws = e._workspace(B)
off = C.c_uint64(); L.check(L.lib.gmvae_workspace_offset(C.byref(d), e.model, b"stamps", C.byref(off)), "off")
raw = ws.view(torch.int64)[off.value // 8: off.value // 8 + nP * 4 * 16].cpu().numpy().reshape(nP * 4, 16).astype(np.float64)
names = ["loads issued", "first half + x landed", "x image + sync", "MFMA 1", "second half landed + sync", "MFMA 2 + publish",
         "exchange polls done", "decoder operands issued, image waited, sync", "bias + ReLU + sync (stage end)"]
cols = [8, 9, 10, 11, 12, 13, 14, 15, 1]
prev = raw[:, 0]
for nm, c in zip(names, cols):
    dlt = raw[:, c] - prev
    print(f"  {nm:46s} +{np.median(dlt):7.0f}  (min {dlt.min():7.0f} max {dlt.max():7.0f})")
    prev = raw[:, c]
print("  stage total", np.median(raw[:, 1] - raw[:, 0]))
