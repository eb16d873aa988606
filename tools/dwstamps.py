"""Wall-clock stamps (100 MHz) of dw_adam's workgroups by tensor (GMVAE_STAMPS=1): where the fused weight-gradient +
optimizer launch spends its time."""
import sys, os, ctypes as C
os.environ.setdefault("GMVAE_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
B = 1024
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
G = 16
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
sx.copy_(x.unsqueeze(0).expand(G, -1, -1))
for _ in range(200): replay()
torch.cuda.synchronize()
d, ws = e._workspace(B)
off = C.c_uint64(); L.check(L.lib.gmvae_workspace_offset(C.byref(d), e.model, b"gstamps", C.byref(off)), "off")
raw = ws.view(torch.int64)[off.value // 8 + 3 * 2048 * 8: off.value // 8 + 4 * 2048 * 8].cpu().numpy().reshape(2048, 8).astype(np.float64)
raw = raw[raw[:, 0] > 0]
t0 = raw[:, 0].min()
print("blocks", len(raw), "launch span %.2f us" % ((raw[:, 4].max() - t0) * 0.01))
names = ["dWy0", "dWg0x", "dWd1", "dWg0y", "dWy1", "dWp", "dWg1", "dWd0"]
for ti in range(8):
    r = raw[(raw[:, 5] == ti) & (raw[:, 4] > 0)]
    if not len(r): continue
    med = lambda a: np.median(a) * 0.01
    print(f"{names[ti]:6s} n={len(r):3d} start {med(r[:,0]-t0):5.2f} | prologue {med(r[:,1]-r[:,0]):5.2f} | contraction {med(r[:,2]-r[:,1]):5.2f} | barrier {med(r[:,3]-r[:,2]):5.2f} | epilogue {med(r[:,4]-r[:,3]):5.2f} | end {med(r[:,4]-t0):5.2f} (max {(r[:,4]-t0).max()*0.01:5.2f})")
r = raw[raw[:, 5] == 99]
if len(r): print(f"tail   n={len(r):3d} start {np.median(r[:,0]-t0)*0.01:5.2f} | end {np.median(r[:,4]-t0)*0.01:5.2f} (max {(r[:,4]-t0).max()*0.01:5.2f})")
