import sys, os, ctypes as C
os.environ["GMVAE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
B = 1024
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
for _ in range(50): e.train_step(x)
torch.cuda.synchronize()
d, ws = e._workspace(B)
off = C.c_uint64(); L.check(L.lib.gmvae_workspace_offset(C.byref(d), e.model, b"stamps", C.byref(off)), "off")
raw = ws.view(torch.int64)[off.value // 8: off.value // 8 + 2 * 64 * 16].cpu().numpy().reshape(2, 64, 16)
names = ["mega (F0 | logits+gumbel | heads | z+hd | decoder | B0 | bwd chain)", "chain_bwd"] if not os.environ.get("GMVAE_NO_MEGA") else ["chain_fwd", "chain_bwd"]
for k, name in enumerate(names):
    st = raw[k][:, :8].astype(np.float64)
    if st.max() == 0: continue
    dur = np.diff(st, axis=1)
    if k == 0 and raw[0][:, 8:12].max() > 0:
        print("  decoder loop segments (sum over chunks) [wait+sync | lambda+epilogue | sync | dhd product]:", np.round(np.median(raw[0][:, 8:12], axis=0)).astype(int), " | issue_dma, mfma_tile, (epilogue = seg1):", np.round(np.median(raw[0][:, 12:14], axis=0)).astype(int))
    print(name, "median per-stage cycles:", np.round(np.median(dur, axis=0)).astype(int), "total", int(np.median(st[:, 7] - st[:, 0])))
