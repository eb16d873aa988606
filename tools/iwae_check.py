import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle as O, hip_util as H
from gmvae_amd.engine import Engine
d = O.Dims(D=3072, L=64, K=64, hidden=(512,), S=50)
p = O.init_params(O.MODEL_GMVAE, d, np.random.default_rng(0))
x, eps, u = O.make_inputs(d, 8)
loss, worst = H.compare_step(O.MODEL_GMVAE, d, p, x, eps, u, grad_rtol=5e-4)   # fp32 ulp of log w ~ -2100 is 2.4e-4: the fp32 oracle itself is 7e-5 off
print("config-5 shapes, B=8: loss", loss, "worst grad rel-to-max err", worst)
e = Engine("gmvae", 3072, 64, 64, [512], n_samples=50, random_seed=0)
B = 512
xb = torch.from_numpy((np.random.default_rng(1).random((B, 3072)) < 0.87).astype(np.uint8)).cuda()
for _ in range(3): e.train_step(xb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): e.train_step(xb)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
fl = O.flops_per_step(O.MODEL_GMVAE, d, B)
print(f"config 5 per-GPU shard B=512 S=50: {dt*1e3:.2f} ms/step, {B*50/dt/1e6:.3f} M ELBO-samples/s, {fl/dt*1e-12:.1f} TFLOP/s (algorithmic)")
