"""Soak of the round-3 / round-4 schedules: train graphs of the skinny schedule (GMVAE at bin/run_train.sh's sizes, VAE, VAE_GMP, a
latent size that is no multiple of 16; the forms above 128 rows: B = 1024 at H = 512 -- the weight-gradient launch whose last-
arriving batch share runs the optimizer -- and B = 2048 at H = 256) and of the general schedule with the plane GEMMs (a small
config-5-shaped model), each for argv[1] seconds: the loss stays finite and falls, no hand-off timeout, and TWO engines started
from the same seed stay bit-identical (a race in a hand-off would show as a difference)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
from gmvae_amd import _lib as L
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
cases = [("gmvae", 784, 128, 10, [512], 64, 1), ("vae", 784, 128, 1, [512], 64, 1), ("vae_gmp", 784, 64, 10, [512], 256, 1),
         ("gmvae", 784, 8, 10, [256], 48, 1), ("gmvae", 1024, 64, 16, [256], 512, 8),
         ("gmvae", 784, 64, 10, [512], 1024, 1), ("gmvae", 784, 128, 10, [256], 2048, 1), ("vae_gmp", 784, 64, 10, [512], 700, 1)]
rng = np.random.default_rng(0)
for model, D, Lz, K, hid, B, S in cases:
    e = Engine(model, D, Lz, K, hid, n_samples=S, random_seed=0)
    e2 = Engine(model, D, Lz, K, hid, n_samples=S, random_seed=0)
    G = 8
    sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
    sx2, replay2 = e2.capture_train_step(B, 1e-3, n_steps=G)
    proto = (rng.random((1, 1, D)) < 0.5)                       # a learnable structure: pixels follow a fixed pattern 90 % of the time
    sx.copy_(torch.from_numpy(((rng.random((G, B, D)) < 0.9) == proto).astype(np.uint8)).cuda())
    sx2.copy_(sx)
    sched = L.step_schedule(e.dims(B), e.model)
    replay(); torch.cuda.synchronize()
    first = e.grads[e.P].item() / B
    t0, n = time.perf_counter(), G
    while time.perf_counter() - t0 < secs:
        for _ in range(20): replay()
        torch.cuda.synchronize()
        n += 20 * G
        loss = e.grads[e.P].item() / B
        assert np.isfinite(loss), (model, n, loss)
        assert e.handoff_timeouts() == 0
    for _ in range(n // G): replay2()                              # the twin: the same number of graph launches
    torch.cuda.synchronize()
    assert torch.equal(e.params, e2.params), (model, "two engines from the same seed differ after", n, "steps")
    assert loss < first, (model, first, loss)
    print(f"{model} D={D} L={Lz} H={hid[0]} B={B} S={S} [{sched}]: {n} steps, loss {first:.2f} -> {loss:.2f}", flush=True)
