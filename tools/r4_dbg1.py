import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle as O
import hip_util as H
name, d, B = "vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(512,)), 530
model = O.MODEL_NAMES[name]
rng = np.random.default_rng(B)
p = O.init_params(model, d, rng)
for k in p:
    if k.endswith("/b"):
        p[k] = rng.normal(0, 0.05, p[k].shape)
x, eps, u = O.make_inputs(d, B, model)
flat = O.pack(model, d, p, np.float32)
lay, _, _ = O.param_layout(model, d)
C = O.loss_and_grads(model, d, p, x, eps, u) if hasattr(O, "loss_and_grads") else None
res = {}
for env in ("skinny", "general"):
    if env == "general": os.environ["GMVAE_NO_SKINNY"] = "1"
    os.environ["GMVAE_NO_MEGA"] = "1"; os.environ["GMVAE_NO_FUSED"] = "1"
    g, t = H.hip_step(model, d, flat, x, eps, u)
    res[env] = g
for nm, shape, off in lay:
    n = int(np.prod(shape))
    a, b = res["skinny"][off:off+n], res["general"][off:off+n]
    print(nm, "skinny vs general rel-to-max", np.abs(a-b).max()/max(np.abs(b).max(),1e-9))
