import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/gmvae_amd") else os.getcwd())
import numpy as np, torch
from gmvae_amd.engine import Engine
def t(model, L, K, H, B, env=None):
    for k in ("GMVAE_NO_SKINNY",): os.environ.pop(k, None)
    if env: os.environ.update(env)
    e = Engine(model, 784, L, K, [H], random_seed=0)
    G = 40
    sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
    sx.copy_(torch.from_numpy((np.random.default_rng(0).random((G, B, 784)) < 0.87).astype(np.uint8)).cuda())
    for _ in range(20): replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (50 * G)
for (m, L, K, H, B) in [("vae_gmp", 64, 10, 512, 256), ("vae_gmp", 128, 10, 512, 64), ("vae", 128, 1, 512, 64), ("vae", 2, 1, 512, 100), ("vae", 16, 1, 512, 100), ("vae", 64, 1, 512, 256), ("vae", 64, 1, 256, 1024)]:
    a = t(m, L, K, H, B)
    b = t(m, L, K, H, B, {"GMVAE_NO_SKINNY": "1"})
    print(f"{m} L={L} H={H} B={B}: skinny-or-default {a:7.2f} us/step | GMVAE_NO_SKINNY {b:7.2f}", flush=True)
