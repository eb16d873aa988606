"""Round-6 soak: (a) the one-launch GMVAE step (mega3_step with every optimizer epilogue behind all flags) for `secs` seconds
against the two-launch form from the same start: parameters and both moments bit-identical at the end, no hand-off timeout;
(b) the same for the VAE family (mega3v_step); (c) the one-launch evaluation replayed between training launches: its bound must
equal a fresh pass's bit for bit each time the parameters stand still, and never produce a non-finite value.
    python tools/soak_round6.py [seconds per part]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rng = np.random.default_rng(0)
for model, Lz, K, B in (("gmvae", 64, 10, 1024), ("vae_gmp", 64, 10, 256), ("vae", 2, 1, 100)):
    G = 40
    xs = torch.from_numpy((rng.random((G, B, 784)) < 0.87).astype(np.uint8)).cuda()
    res, n_launch = [], None
    for fused in (True, False):
        if fused:
            os.environ.pop("GMVAE_NO_FUSE", None)
        else:
            os.environ["GMVAE_NO_FUSE"] = "1"
        e = Engine(model, 784, Lz, K, [64], random_seed=0)
        sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
        sx.copy_(xs)
        t0, n = time.perf_counter(), 0
        while (n_launch is None and time.perf_counter() - t0 < secs) or (n_launch is not None and n < n_launch):
            for _ in range(50):
                replay()
            torch.cuda.synchronize()
            n += 50
            assert e.handoff_timeouts() == 0 and np.isfinite(e.grads[e.P].item()), (model, fused, n)
        if n_launch is None:
            n_launch = n
        res.append((e.params.detach().clone(), e.m.clone(), e.v.clone()))
        print(f"{model} {'one launch' if fused else 'two launches'}: {n * G} steps, {(time.perf_counter() - t0) / (n * G) * 1e6:.2f} us/step incl. host checks", flush=True)
    os.environ.pop("GMVAE_NO_FUSE", None)
    same = all(torch.equal(a, b) for a, b in zip(*res))
    print(f"{model}: parameters and moments after {n_launch * G} steps bit-identical between the two forms: {same}", flush=True)
    assert same
# (c) evaluation between training launches
e = Engine("gmvae", 784, 64, 10, [64], n_samples=50, random_seed=1)
x = torch.from_numpy((rng.random((1024, 784)) < 0.87).astype(np.uint8)).cuda()
sx, replay = e.capture_train_step(1024, 1e-3, n_steps=8)
sx.copy_(x.unsqueeze(0).expand(8, -1, -1))
t0, n = time.perf_counter(), 0
while time.perf_counter() - t0 < secs:
    replay()
    a = e.forward(x)["tail"].clone()
    b = e.forward(x)["tail"].clone()          # (reuses the images)
    c = e.forward(x)["tail"].clone()
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(b, c) and torch.isfinite(a).all(), (n, a, b, c)
    n += 1
print(f"evaluation: {3 * n} passes between {n} training launches: repeated passes bit-identical, all finite; bound {a[0].item() / 1024:.3f}")
