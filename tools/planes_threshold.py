import sys, os, subprocess
code = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gmvae_amd.engine import Engine
D, H, B, S = map(int, sys.argv[1:5])
e = Engine("gmvae", D, 64, 16, [H], n_samples=S, random_seed=0)
G = 4
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
sx.copy_(torch.from_numpy((np.random.default_rng(0).random((G, B, D)) < 0.87).astype(np.uint8)).cuda())
for _ in range(5): replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): replay()
e1.record(); torch.cuda.synchronize()
print(e0.elapsed_time(e1) * 1e3 / (20 * G))
'''
for D, H, B, S in [(1024, 256, 512, 8), (1024, 512, 512, 8), (3072, 512, 512, 8), (3072, 256, 1024, 8), (1024, 512, 2048, 8), (3072, 512, 512, 50)]:
    r = []
    for env in ({}, {"GMVAE_NO_PLANES": "1"}):
        p = subprocess.run([sys.executable, "-c", code, str(D), str(H), str(B), str(S)], env=dict(os.environ, **env), capture_output=True, text=True)
        r.append(float(p.stdout.strip().splitlines()[-1]))
    print(f"D={D} H={H} B={B} S={S} (R={B*S}): planes {r[0]:8.1f} us/step | fp32 {r[1]:8.1f} | ratio {r[1]/r[0]:.2f}", flush=True)
