"""adam_tiles (the data-parallel step's optimizer in tile shape) against the generic adam_tf_img (GMVAE_NO_ADAM_TILES=1) on a
one-rank communicator: parameters and moments must agree bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
bad = 0
for B in (1024, 1000, 256):
    xs = torch.from_numpy((np.random.default_rng(B).random((8, B, 784)) < 0.87).astype(np.uint8)).cuda()
    res = []
    for tiles in (True, False):
        if tiles: os.environ.pop("GMVAE_NO_ADAM_TILES", None)
        else: os.environ["GMVAE_NO_ADAM_TILES"] = "1"
        e = Engine("gmvae", 784, 64, 10, [64], random_seed=7)
        e.enable_rccl()
        sx, replay = e.capture_train_step(B, 1e-3, all_reduce=True, n_steps=8)
        sx.copy_(xs)
        for _ in range(4): replay()
        torch.cuda.synchronize()
        res.append((e.params.detach().clone(), e.m.clone(), e.v.clone(), e.dp_mode, e.handoff_timeouts()))
    os.environ.pop("GMVAE_NO_ADAM_TILES", None)
    same = all(torch.equal(a, b) for a, b in zip(res[0][:3], res[1][:3]))
    print(f"B={B}: adam_tiles == adam_tf_img bit for bit: {same}; mode {res[0][3]}; timeouts {res[0][4]}/{res[1][4]}; finite {bool(torch.isfinite(res[0][0]).all())}")
    bad += not same
sys.exit(1 if bad else 0)
