"""-m gpu: the kernels the benchmark actually times, against the CPU oracle.

bench.py's steady state is an n-step train graph: in-kernel Philox noise, the first layer inside mega_fwd_bwd
(template instance FLT = 1) on weight images that finalize_adam scattered, the bf16x3 weight-gradient launch and
TF-Adam fused into the last launch.  These tests replay such graphs and compare with `oracle.train_step` iterated
over the same batches in fp64, fed with the noise the device drew (gmvae_noise_fill exposes the exact stream:
same function, same counters), for every hand-off variant of the mega schedule: Q = 4 (B <= 1024), Q = 2
(B <= 2048), Q = 1 (B = 8192, the unsharded BASELINE configs[3] batch).

Reference lines reproduced: scripts/gmvae.py:238-267 / scripts/vae.py:167-185 (loss), scripts/runners.py:181-183
(AdamOptimizer.compute_gradients / apply_gradients).  Gates: ELBO of the LAST step <= 1e-4 relative (it is computed
from parameters that went through n-1 device updates), every gradient tensor of the last step <= 1e-4 of its
max, parameters after n TF-Adam steps (see _compare_params)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu

LR = 1e-3


def _noise(L, rows, Lz, K, row_base, seed, step, want_u):
    eps = torch.empty(rows, Lz, dtype=torch.float32, device="cuda")
    u = torch.empty(rows, K, dtype=torch.float32, device="cuda") if want_u else None
    L.check(L.lib.gmvae_noise_fill(L.ptr(eps), L.ptr(u), rows, Lz, K, row_base, seed, step, None, L.current_stream()), "noise")
    torch.cuda.synchronize()
    return eps.cpu().numpy(), (u.cpu().numpy() if want_u else None)


from hip_util import NETS, PRE_TOL      # (the nets' workspace tags; a ReLU may take the other side than in fp64 only where
                                        #  |pre-activation| <= PRE_TOL * sum_k |a_k| |w_kj|: measured over all cases, 6 such
                                        #  units in 5 of 60 trajectories, the largest ratio 5.0e-7)
FLIPS = []             # (case, step, net, |pre| / sum |a||w|) of every unit where the device's ReLU mask differs from fp64's
DRIFT_FLIPS = []       # (case, step, net, units, largest |pre| / sum|a||w|): units the ORACLE's own parameters put on the other side
                       # (not numerically zero anywhere: the oracle keeps its own mask there)


def _device_masks(e, model, d, B):
    """The ReLU masks of the device's LAST step: (kept activation > 0) of every hidden layer, read from the engine's
    workspace (gmvae_workspace_offset "he<i>" / "hg<i>" / "hd<i>"; rows = B for the encoder of x, B*S otherwise)."""
    from gmvae_amd import _lib as L
    cd = e.dims(B)
    ws = e._ws[(B, e.S)]
    out = {}
    for net, tag, per_sample in NETS[model]:
        ms = [None]
        for i, h in enumerate(d.hidden, start=1):
            off = C.c_uint64()
            L.check(L.lib.gmvae_workspace_offset(C.byref(cd), model, f"{tag}{i}".encode(), C.byref(off)), f"offset {tag}{i}")
            rows = B * d.S if per_sample else B
            ms.append(ws[off.value // 4: off.value // 4 + rows * h].view(rows, h).cpu().numpy() > 0)
        out[net] = ms
    return out


def _oracle_trajectory(L, model, d, flat0, xs, seed, step0=0, row_base=0, masks_of_step=None, params_of_step=None, tag=""):
    """n oracle steps (fp64) on the noise of device steps step0 .. step0+n-1; returns (flat, C_last, g_last, [g_t]).

    masks_of_step(t) -> the device's ReLU masks of step t (see _device_masks); params_of_step(t) -> the device's parameters
    BEFORE step t (None: the oracle's own).  ReLU has no derivative at 0 and an fp32 pre-activation that is zero to within rounding can land on the
    other side than the fp64 one: there -- and ONLY there: every unit where the device's mask differs from the fp64 mask AT
    THE DEVICE'S OWN PARAMETERS must have |pre| <= PRE_TOL * sum |a||w| in fp64 (a kernel statement: the same parameters, the
    same noise, only rounding apart) -- the oracle trajectory takes the device's subgradients, so that every parameter seed
    runs inside the same gates instead of a seed being picked that happens to have no such unit."""
    flat = flat0.astype(np.float64)
    m, v = np.zeros_like(flat), np.zeros_like(flat)
    gs = []
    for t in range(xs.shape[0]):
        B = xs[t].shape[0]
        eps, u = _noise(L, B * d.S, d.L, d.K, row_base * d.S, seed, step0 + t, model == O.MODEL_GMVAE)
        masks = None
        if masks_of_step is not None:
            dev_masks = masks_of_step(t)
            at = flat if params_of_step is None else params_of_step(t)
            Cc = O.forward(model, d, O.unpack(model, d, at), xs[t], eps, u, np.float64)
            # The oracle steps with the DEVICE's masks.  A kernel's mask error cannot hide behind that: at the device's own
            # parameters every unit where its mask differs from fp64's is asserted numerically zero (above / below).  What the
            # oracle's own parameters (different by the drift of t Adam steps) would have chosen is only RECORDED
            # (DRIFT_FLIPS, printed by test_trajectory_margins_report): keeping the oracle's own mask there was tried
            # (round 5) and fails for the reason tools/traj_diag.py documents -- one unit on the other side moves the KL
            # term by 5e-4 (vae-L2-H64-B100-seed13), five times the term's gate, in ANY two correct fp32/fp64 trajectories.
            Co = Cc if params_of_step is None else O.forward(model, d, O.unpack(model, d, flat), xs[t], eps, u, np.float64)
            masks = {}
            for net, ms in dev_masks.items():
                out = [None]
                for i in range(1, len(ms)):
                    pre, mag = Cc["pre"][net][i - 1]
                    diff = ms[i] != (pre > 0)
                    if diff.any():
                        ratio = np.abs(pre[diff]) / np.maximum(mag[diff], 1e-30)
                        FLIPS.extend((tag, step0 + t, net, float(r)) for r in ratio)
                        assert ratio.max() <= PRE_TOL, (f"{tag} step {step0 + t} {net} layer {i}: the device's ReLU mask differs from "
                                                        f"fp64's at a pre-activation that is NOT numerically zero "
                                                        f"(|pre| / sum|a||w| = {ratio.max():.2e})")
                    pre_o, mag_o = Co["pre"][net][i - 1]
                    own = pre_o > 0
                    zero_o = np.abs(pre_o) <= PRE_TOL * np.maximum(mag_o, 1e-30)
                    take = (ms[i] != own) & (diff | zero_o)
                    drift = (ms[i] != own) & ~take                # the two trajectories' parameters put this unit on different sides
                    if drift.any():
                        DRIFT_FLIPS.append((tag, step0 + t, net, int(drift.sum()),
                                            float((np.abs(pre_o[drift]) / np.maximum(mag_o[drift], 1e-30)).max())))
                    out.append(ms[i])
                masks[net] = out
        flat, m, v, Cc, g = O.train_step(model, d, flat, m, v, step0 + t + 1, xs[t], eps, u, lr=LR, dtype=np.float64,
                                         relu_masks=masks)
        gs.append(g)
    return flat, Cc, g, gs


def _compare_last_step(model, d, eng, B, Cc, g, at_device=None, tag=""):
    """The LAST step of the trajectory.  Loss terms: against the fp64 oracle trajectory, at the step's gates.  Gradients:
    (a) against the fp64 oracle evaluated AT THE DEVICE's own parameters before that step (`at_device` = (C, g); the
    parameters come from the second engine that steps the same graph kernels one launch at a time and is bit-identical) at
    1e-4 of each tensor's largest entry -- the kernels' error alone, on parameters that went through n - 1 device updates;
    (b) against the oracle TRAJECTORY's last gradient at 2e-3: that difference is the kernels' error PLUS what the n - 1
    steps' Adam drift (_compare_params bounds it) does to the gradient through the loss's curvature -- tools/traj_diag.py:
    VAE L = 2, seed 13: 8.7e-7 at the device's parameters, 2.6e-4 against the trajectory, with the parameters inside their
    tolerance; no fp32 implementation, the reference's included, is pinned tighter than that by an fp64 trajectory."""
    P = eng.P
    buf = eng.grads.cpu().numpy().astype(np.float64)
    tail = buf[P:]
    assert tail[4] == B
    assert abs(tail[0] / B - Cc["loss"]) <= 1e-4 * abs(Cc["loss"]), (tag, tail[0] / B, Cc["loss"])
    assert abs(tail[1] / B - Cc["nll"]) <= 1e-4 * abs(Cc["nll"]), tag
    assert abs(tail[2] / B - Cc["kl"]) <= 1e-4 * max(abs(Cc["kl"]), 1.0), (tag, tail[2] / B, Cc["kl"])       # each term relative to
    assert abs(tail[3] / B - Cc["nent"]) <= 1e-4 * max(abs(Cc["nent"]), 1.0), (tag, tail[3] / B, Cc["nent"])  # itself
    lay, _, _ = O.param_layout(model, d)
    for name, shape, off in lay:
        n = int(np.prod(shape))
        got = buf[off:off + n] / B
        if at_device is not None:
            ref = at_device[1][off:off + n]
            err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)
            GRAD_STATS.append((tag, name, err))
            assert err <= 1e-4, f"{tag} {name}: last-step gradient vs the oracle at the device's parameters, rel-to-max err {err:.2e}"
        ref = g[off:off + n]
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)
        assert err <= (2e-3 if at_device is not None else 1e-4), f"{tag} {name}: last-step gradient vs the oracle trajectory, rel-to-max err {err:.2e}"


GRAD_STATS = []        # (case, tensor, last-step gradient error at the device's parameters / the tensor's largest entry)
import os
GRAD_ERR_GATE = 1e-5   # the device's gradient error at its own parameters (GRAD_STATS, in units of the tensor's largest gradient) must
                       # stay below this for the Adam-derived parameter tolerance to mean anything (measured worst: 3.7e-6)
GRAD_ERR_GRID = (1e-6, 2e-6, 3e-6, 4e-6, 5e-6, 6e-6, 8e-6, 1e-5)
PARAM_STATS = []       # (case, n, {E: (worst |dtheta| / tolerance(E), share of elements constrained to lr / 10 at E, worst tensor)})


def _compare_params(model, d, eng, flat_ref, gs, n, tag=""):
    """Parameters after n device TF-Adam steps vs the fp64 oracle, element by element, with a tolerance that follows from
    Adam itself.  A step moves an element by alpha * m / (sqrt(v) + eps) = O(lr) * sign-like ratio: a gradient error delta
    changes that by about lr * c * |delta| / |g| (c of order 1-3: d/dg of m / sqrt(v)), saturating at ~2.5 lr when |g| is
    itself rounding noise -- in ANY fp32 implementation, the reference's included.  With the device's gradient error E in
    units of the tensor's largest gradient:
        tol_i(E) = 3e-5 + lr * sum_t min(2.5, 3 * E * max|g_t| / |g_t,i|)
    i.e. 3e-5 (3 % of one step) for an element whose gradient is near its tensor's largest, ~ lr only where the gradient is
    itself rounding noise.  E is NOT a constant of this file: the comparison is evaluated on a grid of E and
    test_trajectory_parameters_within_the_measured_gradient_error (below, after every case) gates each trajectory at the
    grid value just above the gradient error this SESSION measured at the device's own parameters (max of GRAD_STATS,
    asserted <= 1e-5).  Here: the loosest grid value (the assertion's bound itself), so that a gross error stops the case."""
    got = eng.params.detach().cpu().numpy().astype(np.float64)
    diff = np.abs(got - flat_ref)
    assert np.isfinite(got).all() and diff.max() <= 2.5 * n * LR
    lay, P, _ = O.param_layout(model, d)
    real = np.zeros(P, bool)
    for name, shape, off in lay:
        real[off:off + int(np.prod(shape))] = True
    ratios = []                                    # per step: max|g_t| / |g_t,i| (inf where the gradient is exactly zero)
    for g in gs:
        r = np.full(P, np.inf)
        for name, shape, off in lay:
            k = int(np.prod(shape))
            ga = np.abs(g[off:off + k])
            # (an element whose gradient is EXACTLY zero -- every weight of a hidden unit whose ReLU is off for the whole batch
            #  -- is not updated by either side: no allowance)
            r[off:off + k] = np.where(ga > 0, max(ga.max(), 1e-30) / np.maximum(ga, 1e-300), np.nan)
        ratios.append(r)
    grid = {}
    for E in GRAD_ERR_GRID:
        tol = np.full(P, 3e-5)
        for r in ratios:
            tol += np.where(np.isnan(r), 0.0, LR * np.minimum(2.5, 3.0 * E * np.nan_to_num(r, nan=0.0)))
        q = diff[real] / tol[real]
        worst_name = ""
        for name, shape, off in lay:
            k = int(np.prod(shape))
            if (diff[off:off + k] / tol[off:off + k]).max() >= q.max():
                worst_name = name
        grid[E] = (float(q.max()), float((tol[real] <= LR / 10).mean()), worst_name)
    PARAM_STATS.append((tag, n, grid))
    w, tight, nm = grid[GRAD_ERR_GRID[-1]]
    assert w <= 1.0, f"{tag} {nm}: |dtheta| / tolerance {w:.2f} even at the assertion's own bound E = {GRAD_ERR_GRID[-1]:.0e}"
    assert tight >= 0.5, f"{tag}: only {tight:.1%} of the parameters are constrained to a tenth of one step"


CASES = [
    # model, D, L, K, hidden, B, n_steps, parameter seeds
    ("gmvae", 784, 64, 10, (64,), 1024, 4, (11, 12)),     # BASELINE configs[2]: Q = 4, steps 2..n in-launch first layer
    ("gmvae", 784, 64, 10, (64,), 1000, 3, (11,)),        # ragged last panel
    ("gmvae", 784, 64, 10, (64,), 2048, 3, (11,)),        # Q = 2
    ("gmvae", 784, 64, 10, (64,), 8192, 3, (11,)),        # Q = 1: the unsharded configs[3] batch on one GPU
    ("vae", 784, 2, 1, (64,), 100, 4, (11, 12, 13)),      # BASELINE configs[0]
    ("vae_gmp", 784, 64, 10, (64,), 256, 4, (11, 12, 13)),  # BASELINE configs[1]
    ("gmvae", 784, 16, 10, (64,), 96, 3, (11,)),          # generic mega instance (not the specialised sizes)
    # bin/run_train.sh sizes, 5 steps, EVERY parameter seed 11..30 (round 3 ran seed 12 only: at 11, 14 and 23 one of the
    # 64 x 512 pre-activations sits within fp32 rounding of zero at some step and its ReLU takes the other side than in
    # fp64; the oracle now follows the device's subgradient at such units and only there, see _oracle_trajectory)
    ("gmvae", 784, 128, 10, (512,), 64, 5, tuple(range(11, 31))),
    ("vae", 784, 8, 1, (96, 96), 48, 3, (11, 12)),        # two hidden layers: general schedule, VAE
    ("vae", 784, 128, 1, (512,), 64, 4, (11, 12, 13, 14)),     # the VAE at bin/run_train.sh's sizes: skinny schedule, eight launches
    ("vae", 784, 32, 1, (256,), 200, 3, (11, 12)),        # skinny VAE, ragged row tiles, in-kernel eps rows over several extra workgroups
    ("gmvae", 784, 8, 10, (256,), 64, 4, (11, 12, 13, 14)),    # skinny, the reference's default latent size (a ragged tile of latent dimensions)
    ("vae_gmp", 784, 64, 10, (512,), 256, 4, (11, 12, 13, 14)),  # skinny VAE_GMP (BASELINE configs[1] at H = 512)
    ("gmvae", 784, 64, 10, (512,), 1024, 3, (11, 12)),    # BASELINE configs[2] at H = 512
    # TF-Adam through sk_dwc's last-arriver finisher (more [64 x 64] tiles than CUs at H = 1024, B >= 512: two batch shares per
    # tile) for the other two models, and the mixture prior's update workgroups riding on that launch
    ("vae", 784, 64, 1, (1024,), 512, 3, (11,)),
    ("vae_gmp", 784, 64, 10, (1024,), 640, 3, (11,)),
    ("vae_gmp", 784, 64, 10, (512,), 768, 3, (11,)),      # (sk_dwb<1>: one workgroup per tile, B > 512)
]


def trajectory_case(model, D, Lz, K, hidden, B, n, seed=11, engine_kw=None):
    """An n-step train graph against n oracle steps.  A second engine with the same seed steps the same batches through n
    launches of a ONE-step graph (the same kernels: the two must agree bit for bit) and hands the oracle the ReLU masks of
    every step."""
    from gmvae_amd import _lib as L
    from gmvae_amd.engine import Engine
    mid = O.MODEL_NAMES[model]
    d = O.Dims(D=D, L=Lz, K=K, hidden=hidden)
    kw = engine_kw or {}
    e = Engine(model, D, Lz, K, list(hidden), random_seed=seed, **kw)
    flat0 = e.params.detach().cpu().numpy()
    xs = (np.random.default_rng(B).random((n, B, D)) < 0.87).astype(np.uint8)
    xd = torch.from_numpy(xs).cuda()
    sx, replay = e.capture_train_step(B, lr=LR, n_steps=n)
    sx.copy_(xd if n > 1 else xd[0])
    replay()
    torch.cuda.synchronize()
    assert e.handoff_timeouts() == 0 and e.global_step == n and int(e.step_dev[0].item()) == n
    e1 = Engine(model, D, Lz, K, list(hidden), random_seed=seed, **kw)
    sx1, replay1 = e1.capture_train_step(B, lr=LR, n_steps=1)
    masks, pre = [], []
    for t in range(n):
        pre.append(e1.params.detach().cpu().numpy().astype(np.float64))         # the device's parameters before step t
        sx1.copy_(xd[t])
        replay1()
        torch.cuda.synchronize()
        masks.append(_device_masks(e1, mid, d, B))
    assert torch.equal(e1.params, e.params), "n launches of a 1-step graph and one launch of an n-step graph must agree bit for bit"
    tag = f"{model}-L{Lz}-H{'x'.join(map(str, hidden))}-B{B}-seed{seed}"
    flat_ref, Cc, g, gs = _oracle_trajectory(L, mid, d, flat0, xs, e.noise_seed, masks_of_step=lambda t: masks[t],
                                             params_of_step=lambda t: pre[t], tag=tag)
    pre_last = pre[n - 1]
    eps, u = _noise(L, B * d.S, d.L, d.K, 0, e.noise_seed, n - 1, mid == O.MODEL_GMVAE)
    C2, g2 = O.loss_and_grads(mid, d, O.unpack(mid, d, pre_last), xs[n - 1], eps, u, np.float64, relu_masks=masks[n - 1])
    _compare_last_step(mid, d, e, B, Cc, g, at_device=(C2, O.pack(mid, d, g2, np.float64)), tag=tag)
    _compare_params(mid, d, e, flat_ref, gs, n, tag)


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}-L{c[2]}-H{c[4][0]}-B{c[5]}-n{c[6]}" for c in CASES])
def test_train_graph_matches_oracle_trajectory(case):
    *shape, seeds = case
    for seed in seeds:
        trajectory_case(*shape, seed=seed)


def test_trajectory_margins_report():
    """Not a gate: prints what the trajectory comparators measured (run after the cases above in file order) -- the units
    where the device's ReLU took the other side than fp64 (with how close to zero the fp64 pre-activation was) and the
    worst parameter difference relative to its Adam-derived tolerance."""
    if FLIPS:
        by = {}
        for tag, step, net, r in FLIPS:
            by.setdefault(tag, []).append(r)
        print(f"\n[trajectory] ReLU units taken from the device: {len(FLIPS)} in {len(by)} trajectories; "
              f"largest |pre| / sum|a||w| {max(r for *_, r in FLIPS):.2e} (gate {PRE_TOL:.0e})")
    if DRIFT_FLIPS:
        print(f"[trajectory] units the oracle's own (drifted) parameters would put on the other side than the device's mask: "
              f"{sum(n for *_, n, _ in DRIFT_FLIPS)} in {len(set(t for t, *_ in DRIFT_FLIPS))} trajectories; largest "
              f"|pre| / sum|a||w| there {max(r for *_, r in DRIFT_FLIPS):.2e}")
    if GRAD_STATS:
        print(f"[trajectory] last-step gradients at the device's parameters: worst rel-to-max error {max(e for *_, e in GRAD_STATS):.2e} (gate 1e-4)")
    if PARAM_STATS and GRAD_STATS:
        E = _session_grad_err()
        print(f"[trajectory] parameters at the session's measured gradient error (grid value {E:.0e}): worst |dtheta| / tolerance "
              f"{max(g[E][0] for *_, g in PARAM_STATS):.2f}; share constrained to lr / 10: min {min(g[E][1] for *_, g in PARAM_STATS):.1%}")


def _session_grad_err():
    """The grid value just above the worst gradient error measured in this session at the device's own parameters."""
    ge = max(e for *_, e in GRAD_STATS)
    assert ge <= GRAD_ERR_GATE, f"device gradient error {ge:.2e} at its own parameters exceeds {GRAD_ERR_GATE:.0e}"
    return min(E for E in GRAD_ERR_GRID if E >= ge)


def test_second_graph_launch_continues_the_trajectory():
    """Two launches of a 3-step graph = 6 oracle steps: the device step counter carries the Philox step and Adam's t
    across launches (sess.run([train_op, global_step]) repeated, scripts/runners.py:231-232)."""
    from gmvae_amd import _lib as L
    from gmvae_amd.engine import Engine
    B, n = 1024, 3
    d = O.Dims(D=784, L=64, K=10, hidden=(64,))
    e = Engine("gmvae", 784, 64, 10, [64], random_seed=4)
    flat0 = e.params.detach().cpu().numpy()
    xs = (np.random.default_rng(5).random((n, B, 784)) < 0.87).astype(np.uint8)
    sx, replay = e.capture_train_step(B, lr=LR, n_steps=n)
    sx.copy_(torch.from_numpy(xs).cuda())
    replay()
    replay()
    torch.cuda.synchronize()
    flat_ref, Cc, g, gs = _oracle_trajectory(L, O.MODEL_GMVAE, d, flat0, np.concatenate([xs, xs]), e.noise_seed)
    _compare_last_step(O.MODEL_GMVAE, d, e, B, Cc, g)
    _compare_params(O.MODEL_GMVAE, d, e, flat_ref, gs, 2 * n, "two-launches")


def test_dp_graph_world1_matches_oracle_trajectory():
    """The data-parallel train graph (RCCL all-reduce captured inside, adam_tf_img after it) with a one-rank
    communicator and a NON-ZERO global row offset: the trajectory of a rank that owns rows [3072, 4096) of a global
    batch, checked against the oracle on the noise rows of exactly those global indices."""
    from gmvae_amd import _lib as L
    from gmvae_amd.engine import Engine
    B, n = 1024, 3
    d = O.Dims(D=784, L=64, K=10, hidden=(64,))
    e = Engine("gmvae", 784, 64, 10, [64], random_seed=8)
    e.rank = 3                                    # as on rank 3 of 8: row0 = 3 * B enters the Philox counters only
    flat0 = e.params.detach().cpu().numpy()
    xs = (np.random.default_rng(6).random((n, B, 784)) < 0.87).astype(np.uint8)
    e.enable_rccl()
    sx, replay = e.capture_train_step(B, lr=LR, all_reduce=True, n_steps=n)
    assert e.dp_mode == "rccl-in-hipgraph"
    sx.copy_(torch.from_numpy(xs).cuda())
    replay()
    torch.cuda.synchronize()
    assert e.handoff_timeouts() == 0
    flat_ref, Cc, g, gs = _oracle_trajectory(L, O.MODEL_GMVAE, d, flat0, xs, e.noise_seed, row_base=3 * B)
    _compare_last_step(O.MODEL_GMVAE, d, e, B, Cc, g)
    _compare_params(O.MODEL_GMVAE, d, e, flat_ref, gs, n, "dp-world1")


EAGER = [
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,)), 1024),               # mega schedule, 4 launches
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,)), 64),               # general schedule (bin/run_train.sh sizes)
    ("gmvae", O.Dims(D=300, L=6, K=7, hidden=(40,), S=3), 24),              # IWAE rows r = b*S + s; L, K not multiples of 4
    ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(64,)), 256),
    ("vae", O.Dims(D=128, L=7, K=1, hidden=(20, 20)), 33),
]


@pytest.mark.parametrize("model,d,B", EAGER, ids=[f"{m}-L{d.L}-S{d.S}-B{B}" for m, d, B in EAGER])
def test_eager_philox_step_matches_oracle(model, d, B):
    """gmvae_step with eps = u = NULL (in-kernel Philox) in every schedule == the oracle on gmvae_noise_fill's arrays,
    including a non-zero row offset (a data-parallel shard)."""
    import hip_util as H
    from gmvae_amd import _lib as L
    mid = O.MODEL_NAMES[model]
    p = O.init_params(mid, d, np.random.default_rng(3))
    flat = O.pack(mid, d, p, np.float32)
    x = (np.random.default_rng(B).random((B, d.D)) < 0.87).astype(np.uint8)
    seed, step, row0 = 0x1234567890AB, 17, 5 * B
    cd = H.dims_of(d, B)
    cd.row0 = row0
    P, _ = L.param_count(cd, mid)
    params, xd = H.dev(flat, torch.float32), H.dev(x, torch.uint8)
    grads = torch.full((P + L.TAIL,), float("nan"), dtype=torch.float32, device="cuda")
    ws = H.workspace(cd, mid)
    L.check(L.lib.gmvae_step(C.byref(cd), mid, L.ptr(xd), None, None, L.ptr(params), L.ptr(grads), L.ptr(ws), seed, step,
                             None, L.current_stream()), "gmvae_step")
    torch.cuda.synchronize()
    eps, u = _noise(L, B * d.S, d.L, d.K, row0 * d.S, seed, step, mid == O.MODEL_GMVAE)
    p32 = O.unpack(mid, d, flat.astype(np.float64))
    Cc, g = O.loss_and_grads(mid, d, p32, x, eps, u, np.float64)
    buf = grads.cpu().numpy().astype(np.float64)
    assert abs(buf[P] / B - Cc["loss"]) <= 1e-4 * abs(Cc["loss"])
    lay, _, _ = O.param_layout(mid, d)
    for name, shape, off in lay:
        n = int(np.prod(shape))
        ref = g[name].ravel()
        err = np.abs(buf[off:off + n] / B - ref).max() / max(np.abs(ref).max(), 1e-6)
        assert err <= 1e-4, f"{name}: {err:.2e}"


def test_pipeline_graph_matches_oracle_from_raw_pixels():
    """The train graph that starts from raw uint8 pixels (binarisation inside the graph): oracle.binarize (bit-exact
    NumPy Philox) -> oracle.train_step on the device's noise, step by step."""
    from gmvae_amd import _lib as L
    from gmvae_amd.data import DeviceDataset
    from gmvae_amd.engine import Engine
    B, n = 1024, 3
    d = O.Dims(D=784, L=64, K=10, hidden=(64,))
    pix = np.random.default_rng(2).integers(0, 256, (5000, 784), dtype=np.uint8)
    ds = DeviceDataset(pix, shuffle=True, seed=3)
    e = Engine("gmvae", 784, 64, 10, [64], random_seed=21)
    flat0 = e.params.detach().cpu().numpy()
    replay = e.capture_train_pipeline(ds, B, lr=LR, n_steps=n)
    replay()
    torch.cuda.synchronize()
    rows = replay.rows.cpu().numpy()
    bseed = e.noise_seed ^ Engine.BINARIZE_SEED_XOR
    xs = np.stack([O.binarize(pix, rows[t], bseed, t) for t in range(n)])
    assert np.array_equal(replay.batches.cpu().numpy(), xs)
    flat_ref, Cc, g, gs = _oracle_trajectory(L, O.MODEL_GMVAE, d, flat0, xs, e.noise_seed)
    _compare_last_step(O.MODEL_GMVAE, d, e, B, Cc, g)
    _compare_params(O.MODEL_GMVAE, d, e, flat_ref, gs, n, "pipeline")


def test_config5_shard_full_size_properties(monkeypatch):
    """BASELINE configs[4] per-GPU shard at FULL size (D = 3072, K = 64, S = 50, H = 512, B = 512: 25,600 sample rows;
    the oracle needs minutes there), through size-independent properties:
      * known answer: all-zero parameters -> loss = D ln 2 - ln K = 2125.189256 for any data and noise, kl = 0, nent = -ln K
        (SURVEY.md section 4), at S = 50 (the IWAE bound of identical samples is the single-sample bound);
      * random parameters: finite loss and gradients, and the gradient SUMS of two half batches (row offsets 0 / 256 in the
        Philox counters) add up to the full batch's -- the data-parallel identity at full size;
      * the top decoder layer's plane GEMMs (bf16 piece products on pre-split operands: what runs at this size) against the
        fp32 MFMA instance on the same step: loss to 1e-6, every gradient tensor to 2e-5 of its maximum;
      * the oracle comparison runs on the first 8 rows with the device's own noise (same kernels, same schedule; 400 sample
        rows: the fp32 instance)."""
    import ctypes as C
    import math
    import hip_util as H
    from gmvae_amd import _lib as L
    d = O.Dims(D=3072, L=64, K=64, hidden=(512,), S=50)
    mid, B = O.MODEL_GMVAE, 512
    lay, P, _ = O.param_layout(mid, d)
    x = (np.random.default_rng(7).random((B, d.D)) < 0.87).astype(np.uint8)

    def run(flat, xs, row0):
        cd = H.dims_of(d, xs.shape[0])
        cd.row0 = row0
        Pp, _ = L.param_count(cd, mid)
        grads = torch.full((Pp + L.TAIL,), float("nan"), dtype=torch.float32, device="cuda")
        ws = H.workspace(cd, mid)
        L.check(L.lib.gmvae_step(C.byref(cd), mid, L.ptr(H.dev(xs, torch.uint8)), None, None, L.ptr(H.dev(flat, torch.float32)),
                                 L.ptr(grads), L.ptr(ws), 5, 2, None, L.current_stream()), "gmvae_step")
        torch.cuda.synchronize()
        return grads.cpu().numpy().astype(np.float64)

    z = run(np.zeros(P, np.float32), x, 0)
    assert z[P] / B == pytest.approx(2125.189256, rel=2e-6)
    assert abs(z[P + 2] / B) < 1e-4 and z[P + 3] / B == pytest.approx(-math.log(64), rel=1e-5) and z[P + 4] == B
    flat = O.pack(mid, d, O.init_params(mid, d, np.random.default_rng(3)), np.float32)
    full = run(flat, x, 0)
    assert np.isfinite(full).all() and 1500 < full[P] / B < 3500
    halves = run(flat, x[:256], 0) + run(flat, x[256:], 256)
    assert abs(halves[P] - full[P]) <= 1e-5 * abs(full[P])
    assert np.abs(halves[:P] - full[:P]).max() <= 1e-4 * np.abs(full[:P]).max()
    assert L.step_schedule(H.dims_of(d, B), mid) == "general+planes"
    monkeypatch.setenv("GMVAE_NO_PLANES", "1")
    f32 = run(flat, x, 0)
    monkeypatch.delenv("GMVAE_NO_PLANES")
    assert not np.array_equal(f32[:P], full[:P])
    assert abs(f32[P] - full[P]) <= 1e-6 * abs(full[P])
    for name, shape, off in lay:
        n = int(np.prod(shape))
        assert np.abs(f32[off:off + n] - full[off:off + n]).max() <= 2e-5 * max(np.abs(f32[off:off + n]).max(), 1e-6), name
    # oracle on the first 8 rows, with the noise the device drew for global rows 0..7
    small = run(flat, x[:8], 0)
    eps, u = _noise(L, 8 * d.S, d.L, d.K, 0, 5, 2, True)
    Cc, g = O.loss_and_grads(mid, d, O.unpack(mid, d, flat.astype(np.float64)), x[:8], eps, u, np.float64)
    assert abs(small[P] / 8 - Cc["loss"]) <= 1e-4 * abs(Cc["loss"])
    for name, shape, off in lay:
        n = int(np.prod(shape))
        ref = g[name].ravel()
        assert np.abs(small[off:off + n] / 8 - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6), name


@pytest.mark.parametrize("B", [1024, 1000, 256])
def test_one_launch_step_equals_two_launch_step_bit_for_bit(B, monkeypatch):
    """mega3_step (csrc/mega3.hpp: the per-row part and the weight-gradient tiles + TF-Adam of scripts/runners.py:181-183,231-232
    in ONE launch, handed over through per-workgroup flags inside the launch) against mega2_fwd_bwd -> dw_adam on the same
    batches: the tiles run the same arithmetic in the same order, so parameters and both Adam moments after 48 steps must agree
    BIT FOR BIT -- a tile that read an operand before it was final, or an update that overtook a reader, shows up here.
    (B = 1024 takes the one-launch form by itself; GMVAE_FUSE=1 forces it for the ragged and the small batch.)"""
    from gmvae_amd.engine import Engine
    from gmvae_amd import _lib as L
    G, n = 16, 48
    rng = np.random.default_rng(B)
    xs = torch.from_numpy((rng.random((G, B, 784)) < 0.87).astype(np.uint8)).cuda()
    res = []
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("GMVAE_NO_FUSE", raising=False); monkeypatch.setenv("GMVAE_FUSE", "1")
        else:
            monkeypatch.setenv("GMVAE_NO_FUSE", "1"); monkeypatch.delenv("GMVAE_FUSE", raising=False)
        e = Engine("gmvae", 784, 64, 10, [64], random_seed=5)
        sx, replay = e.capture_train_step(B, LR, n_steps=G)
        sx.copy_(xs)
        for _ in range(n // G):
            replay()
        torch.cuda.synchronize()
        names = [nm for nm, _, _, _ in e.profile_train_levels(xs[0], lr=LR, iters=2)]
        assert e.handoff_timeouts() == 0 and int(e.step_dev[0].item()) >= n
        res.append((e.params.detach().clone(), e.m.clone(), e.v.clone(), names))
    assert res[0][3] == ["mega3_step"] and res[1][3] == ["mega2_fwd_bwd", "dw_adam"], (res[0][3], res[1][3])
    for a_, b_ in zip(res[0][:3], res[1][:3]):
        assert torch.isfinite(a_).all() and torch.equal(a_, b_)


@pytest.mark.parametrize("model,Lz,K,B", [("vae", 2, 1, 100), ("vae_gmp", 64, 10, 256), ("vae_gmp", 64, 10, 333)])
def test_one_launch_step_of_the_vae_family_equals_two_launch_step_bit_for_bit(model, Lz, K, B, monkeypatch):
    """mega3v_step (csrc/mega3.hpp: mega2v_fwd_bwd's per-row part, the weight-gradient tiles + TF-Adam and -- VAE_GMP -- the mixture
    prior's update blocks in ONE launch of 256 workgroups, most of them without a per-row role) against mega2v_fwd_bwd -> dw_adam
    on the same batches (BASELINE configs[0] / configs[1] and a ragged batch): parameters and both Adam moments after 48 steps
    must agree BIT FOR BIT (scripts/vae.py:153-188, scripts/runners.py:181-183,231-232)."""
    from gmvae_amd.engine import Engine
    G, n = 16, 48
    rng = np.random.default_rng(B)
    xs = torch.from_numpy((rng.random((G, B, 784)) < 0.87).astype(np.uint8)).cuda()
    res = []
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("GMVAE_NO_FUSE", raising=False)
        else:
            monkeypatch.setenv("GMVAE_NO_FUSE", "1")
        e = Engine(model, 784, Lz, K, [64], random_seed=5)
        sx, replay = e.capture_train_step(B, LR, n_steps=G)
        sx.copy_(xs)
        for _ in range(n // G):
            replay()
        torch.cuda.synchronize()
        names = [nm for nm, _, _, _ in e.profile_train_levels(xs[0], lr=LR, iters=2)]
        assert e.handoff_timeouts() == 0 and int(e.step_dev[0].item()) >= n
        res.append((e.params.detach().clone(), e.m.clone(), e.v.clone(), names))
    assert res[0][3] == ["mega3v_step"] and res[1][3] == ["mega2v_fwd_bwd", "dw_adam"], (res[0][3], res[1][3])
    for a_, b_ in zip(res[0][:3], res[1][:3]):
        assert torch.isfinite(a_).all() and torch.equal(a_, b_)


def test_trajectory_parameters_within_the_measured_gradient_error():
    """The gate of every trajectory's parameters (run after the cases above in file order): element-wise |dtheta| within the
    Adam-derived tolerance at the gradient error THIS session measured (max over all cases of the last-step gradient error at
    the device's own parameters -- not a pasted constant), and at least half of the elements constrained to a tenth of a step."""
    if not PARAM_STATS or not GRAD_STATS:
        pytest.skip("no trajectory case ran in this session")
    E = _session_grad_err()
    bad = [(tag, g[E]) for tag, _, g in PARAM_STATS if g[E][0] > 1.0 or g[E][1] < 0.5]
    assert not bad, f"at the measured gradient error (grid {E:.0e}): {bad[:5]}"


@pytest.mark.parametrize("model,Lz,K,B,pnl,name", [("gmvae", 64, 10, 1024, 63, "mega3_step"), ("gmvae", 64, 10, 1024, 0, "mega3_step"),
                                                   ("vae_gmp", 64, 10, 256, 5, "mega3v_step"), ("vae", 2, 1, 100, 6, "mega3v_step")])
def test_one_launch_step_is_all_or_nothing_when_a_lead_gives_up_mid_launch(model, Lz, K, B, pnl, name, monkeypatch):
    """A hand-off that gives up DURING a one-launch step (GMVAE_DEBUG_LEAD_FAULT: one panel's lead sets the error word in front
    of its flag, i.e. after the producers' flags are out and phase P of the tiles has started): the step must not be applied
    anywhere -- parameters, both Adam moments bit-unchanged, NaN loss, the explicit flag raised -- which is what run_train's
    recovery (rewind, safe schedule) relies on (gmvae_amd/runners.py; the two-launch form had the error word final before
    dw_adam started).  Before round 6 the phase-P tiles applied TF-Adam behind the producers' flags alone and the tail slot
    published a finite loss."""
    from gmvae_amd.engine import Engine
    rng = np.random.default_rng(B)
    xs = torch.from_numpy((rng.random((B, 784)) < 0.87).astype(np.uint8)).cuda()
    e = Engine(model, 784, Lz, K, [64], random_seed=5)
    sx, replay = e.capture_train_step(B, LR, n_steps=1)
    sx.copy_(xs)
    for _ in range(3):
        replay()
    torch.cuda.synchronize()
    assert e.handoff_timeouts() == 0 and np.isfinite(replay.tail_log.cpu().numpy()).all()
    p0, m0, v0 = e.params.detach().clone(), e.m.clone(), e.v.clone()
    monkeypatch.setenv("GMVAE_DEBUG_LEAD_FAULT", str(pnl))
    e.drop_graphs()
    sx, replay = e.capture_train_step(B, LR, n_steps=1)
    sx.copy_(xs)
    replay()
    torch.cuda.synchronize()
    assert [nm for nm, *_ in e.profile_train_levels(xs, lr=LR, iters=1)] == [name]
    assert e.handoff_timeouts() == 1
    assert torch.isnan(replay.tail_log[0, 0])
    assert torch.equal(e.params, p0) and torch.equal(e.m, m0) and torch.equal(e.v, v0)
    # the same engine, fault removed: the next step applies
    monkeypatch.delenv("GMVAE_DEBUG_LEAD_FAULT")
    e.drop_graphs()
    sx, replay = e.capture_train_step(B, LR, n_steps=1)
    sx.copy_(xs)
    replay()
    torch.cuda.synchronize()
    assert e.handoff_timeouts() == 0 and torch.isfinite(replay.tail_log[0, 0]) and not torch.equal(e.params, p0)
