import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle as O
import hip_util as H
from gmvae_amd import _lib as L
d = O.Dims(D=200, L=8, K=10, hidden=(64,)); B = 8; model = O.MODEL_GMVAE
p = O.init_params(model, d, np.random.default_rng(1))
x, eps, u = O.make_inputs(d, B)
flat = O.pack(model, d, p, np.float32)
Cc, g = O.loss_and_grads(model, d, O.unpack(model, d, flat.astype(np.float64)), x, eps, u)
cd = H.dims_of(d, B)
P, _ = L.param_count(cd, model)
params, xd, ed, ud = H.dev(flat, torch.float32), H.dev(x, torch.uint8), H.dev(eps, torch.float32), H.dev(u, torch.float32)
grads = torch.zeros(P + 8, dtype=torch.float32, device="cuda"); ws = H.workspace(cd, model)
L.check(L.lib.gmvae_step(C.byref(cd), model, L.ptr(xd), L.ptr(ed), L.ptr(ud), L.ptr(params), L.ptr(grads), L.ptr(ws), 0, 0, None, L.current_stream()), "step")
torch.cuda.synchronize()
def buf(name, shape):
    off = C.c_uint64(); L.check(L.lib.gmvae_workspace_offset(C.byref(cd), model, name.encode(), C.byref(off)), name)
    n = int(np.prod(shape)); return ws[off.value // 4: off.value // 4 + n].cpu().numpy().reshape(shape)
w = 1.0 / B
ref = {"qp": Cc["qp"], "pp": Cc["pp"], "z": Cc["z"], "hd1": Cc["hs_d"][1], "hg1": Cc["hs_g"][1], "hy1": Cc["hs_y"][1], "y": Cc["y"],
       "dqp": Cc["dqp"] / w}
for k, r in ref.items():
    got = buf(k, r.shape)
    print(k, np.abs(got - r).max(), np.abs(r).max())
got = buf("dqp", (B, 16)); r = ref["dqp"]
print(np.round(got[0], 4)); print(np.round(r[0], 4))
pp_ = O.unpack(model, d, flat.astype(np.float64))
dqp_r = ref["dqp"]; hg1 = Cc["hs_g"][1]
dhg = (dqp_r @ pp_["encoder_gmm_fcnet/linear_1/w"].T) * (hg1 > 0)
got = buf("dbuf1", dhg.shape); print("dhg1p", np.abs(got - dhg).max(), np.abs(dhg).max())
lay, P_, _ = O.param_layout(model, d)
offs = {n: (o, s) for n, s, o in lay}
sl = buf("slabs", (P_,))
for nm in ["encoder_gmm_fcnet/linear_1/w", "encoder_gmm_fcnet/linear_1/b", "prior_gmm_fcnet/linear_0/w", "decoder_fcnet/linear_0/w"]:
    o, s = offs[nm]; n = int(np.prod(s))
    r = g[nm] * B
    print(nm, "slab err", np.abs(sl[o:o+n].reshape(s) - r).max(), "ref max", np.abs(r).max())
o, s = offs["encoder_gmm_fcnet/linear_1/w"]
print(np.round(sl[o:o+16], 3)); print(np.round((g["encoder_gmm_fcnet/linear_1/w"] * B)[0], 3))
print(np.round((hg1.T @ dqp_r)[0], 3))
gg = grads.cpu().numpy()
print("grads vs slab maxdiff", np.abs(gg[:P_] - sl).max())
for nm, (o, s) in offs.items():
    n = int(np.prod(s)); r = (g[nm] * B).ravel()
    print(f"{nm:34s} grads err {np.abs(gg[o:o+n] - r).max():.3e} slab err {np.abs(sl[o:o+n] - r).max():.3e}")
