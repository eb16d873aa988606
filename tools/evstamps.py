"""Phase times inside evalf_rows (csrc/evalf.hpp) from its device-clock stamps (GMVAE_EV_STAMPS=1): per workgroup, microseconds
from the launch's first stamp.   python tools/evstamps.py [B] [S]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GMVAE_EV_STAMPS"] = "1"
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
for _ in range(5):
    e.forward(x, n_samples=S)
torch.cuda.synchronize()
d, ws = e._workspace(B, S)
off = C.c_uint64()
L.check(L.lib.gmvae_workspace_offset(C.byref(d), e.model, b"ev_dbg", C.byref(off)), "ev_dbg")
st = ws.view(torch.int64)[off.value // 8: off.value // 8 + 256 * 16].cpu().numpy().reshape(256, 16).astype(np.float64)
t0 = st[:, 0].min()
us = (st - t0) * 0.01
names = {0: "start", 1: "tables done", 2: "round 0 start", 3: "round 0 chain end", 4: "round 0 top end", 6: "round 1 start", 7: "round 1 chain end",
         8: "round 1 top end", 14: "rounds done", 15: "arrived"}
for i, nm in names.items():
    v = us[:, i]
    print(f"{nm:20s} median {np.median(v):7.2f}  min {v.min():7.2f}  max {v.max():7.2f}")
