"""Wall-clock stamps (100 MHz) of mega3_step's workgroups (GMVAE_STAMPS=1): the per-row role's end, then the worker phase by
tensor -- operands requested, all leads' flags seen, contraction done, partial tiles met in LDS, optimizer epilogue done."""
import sys, os, ctypes as C
os.environ.setdefault("GMVAE_M3_STAMPS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd import _lib as L
from gmvae_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
MODEL = sys.argv[2] if len(sys.argv) > 2 else "gmvae"          # gmvae | vae (latent 2) | vae_gmp (latent 64, K = 10)
e = Engine(MODEL, 784, {"gmvae": 64, "vae": 2, "vae_gmp": 64}[MODEL], {"gmvae": 10, "vae": 1, "vae_gmp": 10}[MODEL], [64], random_seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((B, 784)) < 0.87).astype(np.uint8)).cuda()
G = 16
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
sx.copy_(x.unsqueeze(0).expand(G, -1, -1))
for _ in range(200): replay()
torch.cuda.synchronize()
d, ws = e._workspace(B)
off = C.c_uint64(); L.check(L.lib.gmvae_workspace_offset(C.byref(d), e.model, b"gstamps", C.byref(off)), "off")
raw_all = ws.view(torch.int64)[off.value // 8 + 3 * 2048 * 8: off.value // 8 + 4 * 2048 * 8].cpu().numpy().reshape(2048, 8).astype(np.float64)
raw, ext = raw_all[:256], raw_all[256:512]
idx = np.arange(256)
if MODEL == "gmvae":
    nP = ((B + 15) // 16 + 1) & ~1
    lead = idx >= 3 * nP
    extra = np.zeros(256, bool)
else:
    nP = (B + 15) // 16
    lead = (idx >= 6 * nP) & (idx < 7 * nP)
    extra = idx >= 7 * nP
live = raw[:, 7] > 0
lead, extra, ext, raw = lead[live], extra[live], ext[live], raw[live]
t0 = raw[:, 7].min()
us = lambda a: a * 0.01
print("workgroups", len(raw), "timeouts", e.handoff_timeouts(), "launch span %.2f us" % us(raw[:, 5].max() - t0))
prod = ~lead & ~extra
print("per-row role ends: producers median %.2f max %.2f | leads median %.2f max %.2f" % (
    us(np.median(raw[prod, 0] - t0)), us((raw[prod, 0] - t0).max()), us(np.median(raw[lead, 0] - t0)), us((raw[lead, 0] - t0).max())))
names = ["dWy0", "dWg0x", "dWd1", "dWg0y", "dWy1", "dWp", "dWg1", "dWd0"] if MODEL == "gmvae" else ["dWe0", "dWd1", "dWe1", "dWd0", "", "", "", ""]
for who, sel in (("producers", ~lead & ~extra), ("leads", lead), ("role-less", extra)):
    for ti in list(range(8)) + [98, 99]:
        r = raw[sel & (raw[:, 6] == ti) & (raw[:, 5] > 0)]
        if not len(r): continue
        med = lambda a: np.median(a) * 0.01
        nm = "tail" if ti == 99 else ("gmp" if ti == 98 else names[ti])
        print(f"{who:9s} {nm:6s} n={len(r):3d} role end {med(r[:,0]-t0):5.2f} | requested {med(r[:,1]-t0):5.2f} | flags seen {med(r[:,2]-t0):5.2f} (max {us((r[:,2]-t0).max()):5.2f})"
              f" | contraction +{med(r[:,3]-r[:,2]):5.2f} | meet +{med(r[:,4]-r[:,3]):5.2f} | epilogue +{med(r[:,5]-r[:,4]):5.2f} | end {med(r[:,5]-t0):5.2f} (max {us((r[:,5]-t0).max()):5.2f})")

q = lambda a: "median %.2f (min %.2f max %.2f)" % (np.median(a) * 0.01, a.min() * 0.01, a.max() * 0.01)
for who, sel in (("producers", prod), ("leads", lead)):
    print(f"{who}: role end -> stores acknowledged {q(ext[sel, 0] - raw[sel, 0])} | -> barrier passed {q(ext[sel, 1] - ext[sel, 0])} | alpha_t {q(ext[sel, 2] - ext[sel, 1])}")
for who, sel in (("producers", prod), ("leads", lead), ("role-less", extra)):
    ok = sel & (ext[:, 6] > 0) & (ext[:, 7] > 0) & (raw[:, 1] > 0)
    if ok.any(): print(f"{who}: alpha_t done -> slot loop entered {q(ext[ok, 6] - ext[ok, 2])} | -> tile descriptor read {q(ext[ok, 7] - ext[ok, 6])} | -> optimizer operands requested {q(raw[ok, 1] - ext[ok, 7])}")
print("leads' role ends sorted (us):", np.round(np.sort(raw[lead, 0] - t0) * 0.01, 2).tolist())
lf = ext[lead, 1].max()
pf = ext[prod, 1].max()
for ph, nm in ((0, "producers' flags"), (1, "leads' flags")):
    sel = ext[:, 5] == ph
    ref = lf if ph else pf
    if sel.any(): print(f"last wait was for the {nm}: n={int(sel.sum())} seen {q(ext[sel, 3] - t0)}; after the LAST such flag's store {q(ext[sel, 3] - ref)}")
