"""Per-stage s_memtime stamps of mega_fwd_bwd (diagnostic build path GMVAE_STAMPS=1), split by producer / consumer."""
import sys, os, ctypes as C
os.environ.setdefault("GMVAE_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
# argv: [model B L K]  (default: the headline configuration)
MODEL = sys.argv[1] if len(sys.argv) > 1 else "gmvae"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
LZ = int(sys.argv[3]) if len(sys.argv) > 3 else 64
KK = int(sys.argv[4]) if len(sys.argv) > 4 else 10
Q = int(os.environ.get("GMVAE_MEGA_Q", "4"))
nP = (B + 15) // 16
if MODEL == "gmvae" and B <= 1024 and LZ == 64 and KK == 10:
    nP = (nP + 1) & ~1                            # mega2_fwd_bwd's grid covers an even number of panels
e = Engine(MODEL, 784, LZ, KK, [64], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
G = int(os.environ.get("GRAPH_STEPS", "16"))
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)          # the hipGraph the bench replays
sx.copy_(x if G == 1 else x.unsqueeze(0).expand(G, -1, -1))
for _ in range(3000 // G): replay()                      # long enough for the clocks to ramp
torch.cuda.synchronize()
d, ws = e._workspace(B)
off = C.c_uint64(); L.check(L.lib.gmvae_workspace_offset(C.byref(d), e.model, b"stamps", C.byref(off)), "off")
raw = ws.view(torch.int64)[off.value // 8: off.value // 8 + nP * Q * 16].cpu().numpy().reshape(nP * Q, 16)
t0 = raw[:, 0].min()
cons = raw[nP * (Q - 1):]
print("stages: F0 | logits+gumbel | heads | z+hd | decoder | B0(+hand-off) | bwd chain")
st = cons[:, :8].astype(np.float64)
print("consumers: start (rel. to first block)", int(np.median(st[:, 0] - t0)), "per-stage", np.round(np.median(np.diff(st, axis=1), axis=0)).astype(int),
      "total", int(np.median(st[:, 7] - st[:, 0])), "end (rel.)", int(np.max(st[:, 7] - t0)))
if os.environ["GMVAE_STAMPS"] == "4":
    c = cons.astype(np.float64)
    names = ["W dma issued", "x loads issued", "noise drawn", "x image stored", "dma landed + sync", "MFMA", "publish + sync", "image dma + poll done", "F0 end"]
    ts = [8, 9, 10, 11, 12, 13, 14, 15, 1]
    prev = 0
    for nm, t in zip(names, ts):
        print(f"  {nm:24s} +{int(np.median(c[:, t] - c[:, prev])):6d}")
        prev = t
    sys.exit(0)
if os.environ["GMVAE_STAMPS"] in ("2", "3"):
    c = cons.astype(np.float64)
    med = lambda a, b: int(np.median(c[:, a] - c[:, b]))
    print("fine (consumers): F0/FL: issue (FL: +stage loads)", med(8, 0), "| loads+dma return (FL: wait+MFMA+publish)", med(9, 8), "| bias/relu/LDS (FL: noise+exchange+F0)", med(1, 9))
    print("   logits ksplit", med(10, 1), "| gumbel-softmax", med(2, 10), "| hg1+prior heads", med(11, 2), "| q head", med(3, 11),
          "| z/logq/logp", med(12, 3), "| hd", med(4, 12))
    print("   B: dz gemm", med(13, 6), "| dqp elementwise", med(14, 13), "| dhg gemm", med(15, 14), "| dy+softmax bwd+dhy", med(7, 15))
    sys.exit(0)
print("  decoder segments [wait+sync | lambda+epilogue | sync | dhd product | issue | mfma]:", np.round(np.median(cons[:, 8:14], axis=0)).astype(int))
for q in range(1, Q):
    pr = raw[nP * (q - 1): nP * q]
    st = pr[:, :6].astype(np.float64)
    print(f"producers q={q}: start", int(np.median(st[:, 0] - t0)), "per-stage", np.round(np.median(np.diff(st, axis=1), axis=0)).astype(int),
          "decoder end (rel.)", int(np.median(st[:, 5] - t0)))
