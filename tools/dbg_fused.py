import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle as O
import hip_util as H
cases = [("L8 B8 D784", O.Dims(D=784, L=8, K=10, hidden=(64,)), 8), ("L8 B32 D784", O.Dims(D=784, L=8, K=10, hidden=(64,)), 32),
         ("L16 B8 D784", O.Dims(D=784, L=16, K=10, hidden=(64,)), 8), ("L8 B8 D200", O.Dims(D=200, L=8, K=10, hidden=(64,)), 8),
         ("L64 B8 D784", O.Dims(D=784, L=64, K=10, hidden=(64,)), 8)]
for name, d, B in cases:
    model = O.MODEL_GMVAE
    p = O.init_params(model, d, np.random.default_rng(1))
    x, eps, u = O.make_inputs(d, B)
    flat = O.pack(model, d, p, np.float32)
    C, g = O.loss_and_grads(model, d, O.unpack(model, d, flat.astype(np.float64)), x, eps, u)
    gs, tail = H.hip_step(model, d, flat, x, eps, u)
    lay, P, _ = O.param_layout(model, d)
    print(name, "loss", tail[0] / B, C["loss"])
    for nm, shape, off in lay:
        n = int(np.prod(shape)); ref = g[nm].ravel(); got = gs[off:off + n] / B
        print(f"   {nm:34s} err {np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-9):.2e}")
