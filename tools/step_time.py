"""us/step of the headline configuration (or argv-selected sizes) on the 40-step train graph, HIP events around each graph
launch; also the per-launch spans / timeline shares.  GMVAE_HIP_LIB selects the library build (A/B runs: tools/ab_step.py).
argv: [config] [seconds]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.argv = sys.argv[:1] + sys.argv[1:]
import bench
from gmvae_amd.engine import Engine
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "configs2"]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
G = int(os.environ.get("GRAPH_STEPS", "40"))
B, D = cfg["batch"], cfg["data_dim"]
if len(sys.argv) > 3: B = int(sys.argv[3])
e = Engine(cfg["model"], D, cfg["latent"], cfg["components"] if cfg["model"] != "vae" else 1, [cfg["hidden"]] * cfg["layers"],
           n_samples=cfg["n_samples"], random_seed=0)
rng = np.random.default_rng(0)
sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
sx.copy_(torch.from_numpy((rng.random((G, B, D)) < 0.87).astype(np.uint8)).cuda())
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.7:
    replay()
torch.cuda.synchronize()
ms = []
t0 = time.perf_counter()
while time.perf_counter() - t0 < secs:
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in evs:
        a.record(); replay(); b.record()
    torch.cuda.synchronize()
    ms += [a.elapsed_time(b) / G for a, b in evs]
ms.sort()
out = {"us_per_step_median": ms[len(ms) // 2] * 1e3, "us_per_step_min": ms[0] * 1e3, "n": len(ms), "timeouts": e.handoff_timeouts(),
       "loss": e.grads[e.P].item() / B}
try:
    x = sx[0] if G > 1 else sx
    out["levels"] = [[n, round(s, 2), round(t, 2)] for n, s, _, t in e.profile_train_levels(x, iters=20)]
except Exception as ex:
    out["levels"] = str(ex)[:80]
print(json.dumps(out))
