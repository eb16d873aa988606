"""Fixed cost of ONE timed region of the driver's command (bench.py --steps 20 = one launch of a 20-step train graph, host clock
around launch + torch.cuda.synchronize): how much of it is the wait itself.  Compares, per region, (a) torch.cuda.synchronize()
alone, (b) a host spin on an event query in front of it, (c) stream.synchronize(); and the same for regions of 20 x 20 steps.
    python tools/sync_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
B, G = 1024, 20
e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
xs, replay = e.capture_train_step(B, 1e-3, n_steps=G)
xs.copy_(torch.from_numpy((np.random.default_rng(0).random((G, B, 784)) < 0.87).astype(np.uint8)))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.75:
    replay(); replay(); torch.cuda.synchronize()
st = torch.cuda.current_stream()


def region(n_launch, how):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n_launch):
        replay()
    if how == "spin":
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
    elif how == "stream":
        st.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e6 / (n_launch * G)


for n in (1, 20):
    for how in ("device", "spin", "stream"):
        v = sorted(region(n, how) for _ in range(30))
        print(f"{n:3d} launch(es) of {G} steps, wait = {how:6s}: median {v[15]:.2f} us/step, min {v[0]:.2f}, max {v[-1]:.2f}", flush=True)
