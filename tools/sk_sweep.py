"""Skinny vs general schedule over the batch size at hidden 512 / latent 128 (us per step, 16-step train graphs)."""
import sys, os, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, time
sys.path.insert(0, %r)
import numpy as np, torch
from gmvae_amd.engine import Engine
B = int(sys.argv[1]); H = int(sys.argv[2])
e = Engine("gmvae", 784, 128, 10, [H], random_seed=0)
G = 16
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
sx.copy_(torch.from_numpy((np.random.default_rng(0).random((G, B, 784)) < 0.87).astype(np.uint8)).cuda())
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5: replay()
torch.cuda.synchronize()
ms = []
for _ in range(5):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); replay(); b.record()
    torch.cuda.synchronize()
    ms += [a.elapsed_time(b) / G for a, b in evs]
ms.sort()
print(ms[len(ms) // 2] * 1e3)
''' % root
for H in [int(h) for h in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("512", "256"))]:
    for B in [int(b) for b in (sys.argv[2].split(",") if len(sys.argv) > 2 else "32,64,128,256,512,1024".split(","))]:
        row = []
        for env in ({"GMVAE_SKINNY_MAXB": "4096"}, {"GMVAE_NO_SKINNY": "1"}):
            p = subprocess.run([sys.executable, "-c", code, str(B), str(H)], env=dict(os.environ, **env), capture_output=True, text=True)
            try:
                row.append(float(p.stdout.strip().splitlines()[-1]))
            except Exception:
                row.append(float("nan")); print(p.stderr[-300:])
        print(f"H={H} B={B:5d}: skinny {row[0]:8.1f} us   general {row[1]:8.1f} us   ratio {row[1] / row[0]:.2f}", flush=True)
