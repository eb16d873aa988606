"""mega3_step (one launch per step) against mega2_fwd_bwd + dw_adam (GMVAE_NO_FUSE=1) on the same batches: the tiles run the
same arithmetic in the same order, so parameters, moments and per-step losses must agree BIT FOR BIT -- a race (a tile reading
an operand before it is final, an update overtaking a reader) shows up as a difference.  argv: [steps] [batch sizes ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmvae_amd.engine import Engine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sizes = [int(v) for v in sys.argv[2:]] or [1024, 1000, 512, 256, 104, 16]
G = 16
bad = 0
for B in sizes:
    rng = np.random.default_rng(B)
    xs = torch.from_numpy((rng.random((G, B, 784)) < 0.87).astype(np.uint8)).cuda()
    res = []
    for fused in (True, False):
        if fused: os.environ.pop("GMVAE_NO_FUSE", None); os.environ["GMVAE_FUSE"] = "1"
        else: os.environ["GMVAE_NO_FUSE"] = "1"; os.environ.pop("GMVAE_FUSE", None)
        e = Engine("gmvae", 784, 64, 10, [64], random_seed=5)
        sx, replay = e.capture_train_step(B, 1e-3, n_steps=G)
        sx.copy_(xs)
        losses = []
        for _ in range(steps // G):
            replay()
            losses.append(e.last_tail_log().clone() if hasattr(e, "last_tail_log") else None)
        torch.cuda.synchronize()
        res.append((e.params.detach().clone(), e.m.clone(), e.v.clone(), e.handoff_timeouts(), e.step_schedule(B) if hasattr(e, "step_schedule") else ""))
    os.environ.pop("GMVAE_NO_FUSE", None); os.environ.pop("GMVAE_FUSE", None)
    same = all(torch.equal(a, b) for a, b in zip(res[0][:3], res[1][:3]))
    d = (res[0][0] - res[1][0]).abs().max().item()
    print(f"B={B:5d} steps={steps // G * G}: fused == two-launch bit for bit: {same} (max |dtheta| {d:.3e}) timeouts {res[0][3]} / {res[1][3]} finite {bool(torch.isfinite(res[0][0]).all())}", flush=True)
    bad += (not same) or res[0][3] != 0
sys.exit(1 if bad else 0)
