"""Soak: many launches of the 40-step train graph (and of the pipeline graph); the hand-off error word must stay
clear and the loss finite.  argv[1] = seconds per variant."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
from gmvae_amd.data import DeviceDataset
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
B, G = 1024, 40
rng = np.random.default_rng(0)
for variant in ("resident batches", "pipeline"):
    e = Engine("gmvae", 784, 64, 10, [64], random_seed=0)
    if variant == "pipeline":
        ds = DeviceDataset(rng.integers(0, 256, (60000, 784), dtype=np.uint8), shuffle=True, seed=3)
        replay = e.capture_train_pipeline(ds, B, 1e-3, n_steps=G)
    else:
        sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
        sx.copy_(torch.from_numpy((rng.random((G, B, 784)) < 0.87).astype(np.uint8)).cuda())
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50): replay()
        torch.cuda.synchronize()
        n += 50 * G
        loss = e.grads[e.P].item() / B
        assert np.isfinite(loss), (variant, n, loss)
        assert e.handoff_timeouts() == 0, (variant, n)
    print(f"{variant}: {n} steps in {time.perf_counter() - t0:.1f} s ({(time.perf_counter() - t0) / n * 1e6:.1f} us/step incl. host checks), loss {loss:.3f}, no hand-off timeout")
