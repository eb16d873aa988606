"""Same inputs twice through the general schedule's plane paths (training step at a config-5-like shape, forward-only pass at the
eval_iwae shape): outputs must agree BIT FOR BIT -- no atomics, no order that depends on timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle as O
import hip_util as H
bad = 0
d = O.Dims(D=3072, L=64, K=64, hidden=(512,), S=50)
rng = np.random.default_rng(3)
p = O.init_params(O.MODEL_GMVAE, d, rng)
flat = O.pack(O.MODEL_GMVAE, d, p, np.float32)
x, eps, u = O.make_inputs(d, 128)
g = [H.hip_step(O.MODEL_GMVAE, d, flat, x, eps, u)[0] for _ in range(3)]
same = all(np.array_equal(g[0], gi) for gi in g[1:])
print("training step, D=3072 H=512 S=50 B=128 (general+planes, f16 pairs): three runs bit-identical:", same, flush=True)
bad += not same
d = O.Dims(D=784, L=64, K=10, hidden=(64,), S=50)
p = O.init_params(O.MODEL_GMVAE, d, rng)
flat = O.pack(O.MODEL_GMVAE, d, p, np.float32)
x, eps, u = O.make_inputs(d, 256)
r = [H.hip_forward(O.MODEL_GMVAE, d, flat, x, eps, u)[1] for _ in range(3)]
same = all(np.array_equal(r[0], ri) for ri in r[1:])
print("forward-only pass, D=784 H=64 S=50 B=256 (forward pairs, rows_small_k): three runs bit-identical:", same, flush=True)
bad += not same
sys.exit(1 if bad else 0)
