"""Wall-clock (s_memrealtime, 100 MHz, one clock for all XCDs) stamps around mega2_fwd_bwd's two in-launch hand-offs
(diagnostic build path GMVAE_STAMPS=5): is a wait the producers' skew or the visibility latency of their stores?"""
import sys, os, ctypes as C
os.environ.setdefault("GMVAE_STAMPS", "5")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
B, Q = 1024, 4
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
raw = ws.view(torch.int64)[off.value // 8: off.value // 8 + nP * Q * 16].cpu().numpy().reshape(nP * Q, 16).astype(np.float64)
w = raw[:, 8:16] * 0.01                                   # microseconds
t0 = w[:, 5].min()
prod = w[:nP * 3].reshape(3, nP, 8)                       # [q-1][panel]
cons = w[nP * 3:]
us = lambda a: f"{np.median(a):7.2f} (min {a.min():6.2f} max {a.max():6.2f})"
print("FL exchange: publish time after the grid's first publish, per quarter:", us(cons[:, 5] - t0), [us(prod[i, :, 5] - t0) for i in range(3)])
last_pub = np.maximum(cons[:, 5], prod[:, :, 5].max(0))
print("  consumer: own publish -> polls done", us(cons[:, 6] - cons[:, 5]), "| panel's LAST publish -> polls done", us(cons[:, 6] - last_pub))
print("D hand-off: producers' D end -> publish issued", us((prod[:, :, 1] - prod[:, :, 0]).ravel()))
lastp = prod[:, :, 1].max(0)
print("  consumer D end relative to the panel's last producer publish", us(cons[:, 0] - lastp))
print("  consumer: D end -> poll start", us(cons[:, 1] - cons[:, 0]), "| poll", us(cons[:, 2] - cons[:, 1]), "| dhd written", us(cons[:, 3] - cons[:, 2]),
      "| dma_wait + barrier", us(cons[:, 4] - cons[:, 3]))
print("  panel's last producer publish -> consumer polls done", us(cons[:, 2] - lastp))
print("  failed sweeps (thread 0): D hand-off, consumers: median", np.median(raw[nP * 3:, 15]), "max", raw[nP * 3:, 15].max(),
      "| first-layer exchange, all workgroups: median", np.median(raw[:, 14]), "max", raw[:, 14].max())
xcc = raw[:, 0].astype(int)
print('XCC ids of workgroups 0..31:', xcc[:32].tolist(), '| leads 192..207:', xcc[192:208].tolist())
bid = np.arange(nP * Q)
print("XCD of a workgroup == blockIdx % 8 for", int((xcc == bid % 8).sum()), "of", len(bid), "workgroups; a panel's four workgroups on one XCD in",
      int(sum(len(set(xcc[[q * nP + p for q in range(3)] + [3 * nP + p]])) == 1 for p in range(nP))), "of", nP, "panels")
