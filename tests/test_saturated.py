"""-m gpu: every fast path OUTSIDE the Xavier regime, against the fp64 oracle.

The per-row chains of the one-launch / skinny / chain kernels take their transcendentals as the bare hardware forms
(v_exp_f32 / v_log_f32 / v_rcp_f32: csrc/gemm.hpp flog / fexp / softplus_sig).  The other parity files initialise Xavier weights
and N(0, 0.05) biases, i.e. logits of O(1) and sigma ~ 1.  Here the same kernels run where a TRAINED model lives and beyond:
  * decoder logits up to |lambda| ~ 60 (a trained MNIST decoder saturates its background pixels), y logits of +-15;
  * raw_sigma spanning [-8, 8] and [-20, 20] per latent dimension in the q head and the prior head (sigma from 2e-9 to 20;
    scripts/base.py:70 `tf.maximum(tf.nn.softplus(raw + bias), sigma_min)`), mixture-prior scales likewise;
  * the factories' own hyper-parameter defaults sigma_min = 0.001, raw_sigma_bias = 0.25 (scripts/vae.py:197-198,
    scripts/gmvae.py:283-286: the runner overrides them with 0.0 / 0.5, a user of create_gmvae() gets these);
  * the uniform stream's two extreme values, u = tiny and u = 1 - 2^-24 (Gumbel noise -4.47 and +16.6): injected
    where a path takes external noise, and REACHED THROUGH THE BATCH'S ROW OFFSET where the kernel draws its own Philox stream
    (tools/find_extreme_u.py found the global rows; the oracle's restatement of the stream is asserted to hold the value);
  * parameters after 304 training steps on structured pixels from DeviceDataset.
Gates: the step's own (ELBO 1e-4 relative, each term relative to itself, every gradient tensor 1e-4 of its max) against
oracle.loss_and_grads at the device's own parameters, on the noise the device drew."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
LR = 1e-3
U_MAX = float(np.nextafter(np.float32(1), np.float32(0)))      # 1 - 2^-24: the largest value of the uniform stream
# tools/find_extreme_u.py 11 0 23: global rows of the in-kernel uniform stream (seed 11, step 0, K = 10) that hold its extremes
SEED = 11
ROW_TINY, K_TINY = 396610, 9
ROW_MAX, K_MAX = 261784, 3


def saturate(model, d, p, rng, x, lam=60.0, span=8.0, logit=15.0):
    """Xavier parameters pushed where a trained model (and a diverging one) lives: raw_sigma biases of the q head and the prior
    head spread over [-span, span] per latent dimension, encoder_y's logits scaled to +-logit, the decoder's output layer scaled
    so that the largest |lambda| on this batch is `lam` (measured with the fp64 oracle on standard noise)."""
    nl = len(d.hidden)
    p = {k: np.array(v, np.float64) for k, v in p.items()}
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    raws = lambda: rng.permutation(np.linspace(-span, span, d.L))
    enc = "encoder_gmm" if model == O.MODEL_GMVAE else "encoder"
    p[f"{enc}_fcnet/linear_{nl}/b"][d.L:] = raws() - d.raw_sigma_bias
    if model == O.MODEL_GMVAE:
        p["prior_gmm_fcnet/linear_0/b"][d.L:] = raws() - d.raw_sigma_bias
        p[f"encoder_y_fcnet/linear_{nl}/w"] *= logit / 1.5
    if model == O.MODEL_VAE_GMP:
        p["raw_scale_diag"] = np.stack([raws() for _ in range(d.K)])
        p["mixture_logits"] = rng.normal(0, 4.0, d.K)
        p["loc"] = rng.normal(0, 2.0, (d.K, d.L))
    B = x.shape[0]
    rn = np.random.default_rng(1)
    eps = rn.standard_normal((B * d.S, d.L))
    u = rn.uniform(1e-6, 1 - 1e-6, (B * d.S, d.K))
    m = np.abs(O.forward(model, d, p, x, eps, u if model == O.MODEL_GMVAE else None)["lam"]).max()
    for s in ("w", "b"):
        p[f"decoder_fcnet/linear_{nl}/{s}"] *= lam / m
    return p


def _check_terms(got, Cc, B, tag):
    loss, nll, kl, nent = (got[i] / B for i in range(4))
    assert got[4] == B, tag
    assert np.isfinite(got[:4]).all(), (tag, got[:5])
    assert abs(loss - Cc["loss"]) <= 1e-4 * abs(Cc["loss"]), (tag, loss, Cc["loss"])
    assert abs(nll - Cc["nll"]) <= 1e-4 * abs(Cc["nll"]), (tag, nll, Cc["nll"])
    assert abs(kl - Cc["kl"]) <= 1e-4 * max(abs(Cc["kl"]), 1.0), (tag, kl, Cc["kl"])
    assert abs(nent - Cc["nent"]) <= 1e-4 * max(abs(Cc["nent"]), 1.0), (tag, nent, Cc["nent"])


def _grad_errs(model, d, gs, g, B):
    lay, _, _ = O.param_layout(model, d)
    out = []
    for name, shape, off in lay:
        n = int(np.prod(shape))
        ref = np.asarray(g[name]).ravel()
        out.append((name, np.abs(gs[off:off + n] / B - ref).max() / max(np.abs(ref).max(), 1e-6)))
    return out


def graph_step_case(model, D, Lz, K, hidden, B, launches, row=None, k_of_row=None, want=None, span=8.0, lam=60.0, hp=None,
                    params=None, step0=0, seed=SEED, x=None):
    """ONE step of a one-step train graph (the kernels bench.py times: in-kernel Philox noise, TF-Adam inside) at saturated
    parameters, against oracle.loss_and_grads at the same parameters on the noise the device drew.  `row`: a global row of the
    uniform stream that must lie inside the batch (the engine's row offset is set to the multiple of B below it)."""
    import hip_util as H
    from test_timed_path import _device_masks, _noise
    from gmvae_amd import _lib as L
    from gmvae_amd.engine import Engine
    mid = O.MODEL_NAMES[model]
    hp = hp or {}
    d = O.Dims(D=D, L=Lz, K=K, hidden=hidden, **hp)
    e = Engine(model, D, Lz, K, list(hidden), random_seed=seed, **hp)
    rng = np.random.default_rng(B + Lz)
    if x is None:
        x = (rng.random((B, D)) < 0.87).astype(np.uint8)
    if params is None:
        p0 = O.unpack(mid, d, e.params.detach().cpu().numpy().astype(np.float64))
        flat = O.pack(mid, d, saturate(mid, d, p0, rng, x, lam=lam, span=span), np.float32)
    else:
        flat = params
    with torch.no_grad():
        e.params.copy_(torch.from_numpy(flat).cuda())
    if row is not None:
        e.rank = row // B                                   # GmvaeDims.row0 = rank * B: the Philox counters hold the global row
    e.global_step = step0
    e.step_dev.fill_(step0)
    row0 = e.rank * B
    eps, u = _noise(L, B * d.S, d.L, d.K, row0 * d.S, e.noise_seed, step0, mid == O.MODEL_GMVAE)
    if row is not None:
        assert u[row - row0, k_of_row] == want, (u[row - row0], want)
        eo, uo = O.noise(B, d.L, d.K, row0, e.noise_seed, step0)
        assert np.array_equal(uo, u)                        # the CPU restatement of the stream holds the same extreme
    sx, replay = e.capture_train_step(B, lr=LR, n_steps=1)
    sx.copy_(torch.from_numpy(x).cuda())
    replay()
    torch.cuda.synchronize()
    assert e.handoff_timeouts() == 0
    got = e.grads.cpu().numpy().astype(np.float64)
    masks = _device_masks(e, mid, d, B)
    tag = f"{model}-L{Lz}-H{hidden[0]}-B{B}-span{span}-{hp}"
    p32 = O.unpack(mid, d, flat.astype(np.float64))
    Cc, g = O.loss_and_grads(mid, d, p32, x, eps, u, np.float64)
    _check_terms(got[e.P:], Cc, B, tag)
    errs = _grad_errs(mid, d, got, g, B)
    if max(err for _, err in errs) > 1e-4 and H.check_masks(masks, Cc["pre"], tag):
        _, g = O.loss_and_grads(mid, d, p32, x, eps, u, np.float64, relu_masks=masks)
        errs = _grad_errs(mid, d, got, g, B)
    for name, err in errs:
        assert err <= 1e-4, f"{tag} {name}: rel-to-max err {err:.3e}"
    assert torch.isfinite(e.params).all()
    if launches is not None:
        names = [nm for nm, *_ in e.profile_train_levels(torch.from_numpy(x).cuda(), lr=LR, iters=2)]
        assert names == launches, names
    return Cc, max(err for _, err in errs)


M3 = ["mega3_step"]
M2 = ["mega2_fwd_bwd", "dw_adam"]
M3V = ["mega3v_step"]
FACTORY = dict(sigma_min=0.001, raw_sigma_bias=0.25)        # scripts/vae.py:197-198, scripts/gmvae.py:283-286


@pytest.mark.parametrize("span,hp,row", [(8.0, None, ("tiny",)), (8.0, None, ("max",)), (20.0, None, None), (8.0, FACTORY, None),
                                         (20.0, FACTORY, ("tiny",))], ids=["span8-u_tiny", "span8-u_max", "span20", "factory", "factory-span20-u_tiny"])
def test_mega3_step_saturated(span, hp, row):
    r = None if row is None else ((ROW_TINY, K_TINY, O.TINY_F32) if row[0] == "tiny" else (ROW_MAX, K_MAX, U_MAX))
    kw = {} if r is None else dict(row=r[0], k_of_row=r[1], want=np.float32(r[2]))
    graph_step_case("gmvae", 784, 64, 10, (64,), 1024, M3, span=span, hp=hp, **kw)


@pytest.mark.parametrize("span,hp,row", [(8.0, None, ("tiny",)), (20.0, FACTORY, ("max",))], ids=["span8-u_tiny", "factory-span20-u_max"])
def test_mega2_two_launch_step_saturated(span, hp, row):
    r = (ROW_TINY, K_TINY, O.TINY_F32) if row[0] == "tiny" else (ROW_MAX, K_MAX, U_MAX)
    graph_step_case("gmvae", 784, 64, 10, (64,), 512, M2, span=span, hp=hp, row=r[0], k_of_row=r[1], want=np.float32(r[2]))


@pytest.mark.parametrize("model,Lz,K,B", [("vae", 2, 1, 100), ("vae_gmp", 64, 10, 256)])
@pytest.mark.parametrize("span,hp", [(8.0, None), (20.0, None), (8.0, FACTORY)], ids=["span8", "span20", "factory"])
def test_mega3v_step_saturated(model, Lz, K, B, span, hp):
    graph_step_case(model, 784, Lz, K, (64,), B, M3V, span=span, hp=hp)


SK_G = ["sk_first_layers", "sk_y_path", "sk_q_head_z", "sk_dec_hidden", "sk_dec_bernoulli", "sk_bwd_dhd", "sk_bwd_dz_heads",
        "sk_bwd_dhg", "sk_y_path_bwd", "sk_dw_adam"]


@pytest.mark.parametrize("Lz,B,span,hp,row", [(128, 64, 8.0, None, "tiny"), (128, 64, 20.0, FACTORY, "max"), (64, 1024, 8.0, None, "max"),
                                              (64, 1024, 20.0, FACTORY, "tiny")])
def test_skinny_step_saturated(Lz, B, span, hp, row):
    """bin/run_train.sh sizes (H = 512, latent 128, batch 64) and BASELINE configs[2] at H = 512."""
    from gmvae_amd import _lib as L
    import hip_util as H
    r = (ROW_TINY, K_TINY, O.TINY_F32) if row == "tiny" else (ROW_MAX, K_MAX, U_MAX)
    d = O.Dims(D=784, L=Lz, K=10, hidden=(512,))
    assert L.step_schedule(H.dims_of(d, B), O.MODEL_GMVAE) == "skinny"
    graph_step_case("gmvae", 784, Lz, 10, (512,), B, None, span=span, hp=hp, row=r[0], k_of_row=r[1], want=np.float32(r[2]))


# ------------------------------------------------------------------ external noise: gmvae_step in every schedule, both extremes injected
EAGER = [
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,)), 1024, "mega2"),
    ("gmvae", O.Dims(D=784, L=16, K=10, hidden=(64,)), 96, "mega"),
    ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(64,)), 256, "mega2v"),
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,)), 64, "skinny"),
    ("vae", O.Dims(D=784, L=32, K=1, hidden=(256,)), 200, "skinny"),
    ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(512,)), 256, "skinny"),
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(48,), gen_bias_init=np.linspace(-3, 3, 784)), 128, "fused"),   # the chain kernels
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,), gen_bias_init=np.linspace(-3, 3, 784)), 128, "skinny"),
    ("gmvae", O.Dims(D=300, L=24, K=7, hidden=(96, 40), S=3), 48, "general"),
    ("vae_gmp", O.Dims(D=256, L=30, K=16, hidden=(128,), S=2), 64, "general"),
]


@pytest.mark.parametrize("span", [8.0, 20.0])
@pytest.mark.parametrize("factory", [False, True], ids=["runner-hp", "factory-hp"])
@pytest.mark.parametrize("name,d,B,sched", EAGER, ids=[f"{n}-{s}-L{d.L}-H{d.hidden[0]}-B{B}" for n, d, B, s in EAGER])
def test_eager_step_saturated(name, d, B, sched, factory, span):
    import dataclasses
    import hip_util as H
    from gmvae_amd import _lib as L
    model = O.MODEL_NAMES[name]
    if factory:
        d = dataclasses.replace(d, **FACTORY)
    assert L.step_schedule(H.dims_of(d, B), model) == sched
    rng = np.random.default_rng(B + d.L)
    x, eps, u = O.make_inputs(d, B, model)
    p = saturate(model, d, O.init_params(model, d, rng), rng, x, span=span)
    if u is not None:
        u[0, 0], u[1, d.K - 1], u[2, :] = O.TINY_F32, U_MAX, U_MAX
        u[3, :] = O.TINY_F32
        u[B * d.S - 1, 1] = O.TINY_F32
    H.compare_step(model, d, p, x, eps, u)


FWD = [("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,), S=50), 256, "evalf"),   # R = 12,800: the one-launch evaluation (csrc/evalf.hpp)
       ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,), S=3), 100, "evalf"),    # ragged panels: 300 sample rows over 100 workgroups
       ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,)), 512, "evalf"),         # S = 1
       ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,), S=50, gen_bias_init=0.7, temperature=0.6), 256, "evalf"),
       ("gmvae", O.Dims(D=784, L=32, K=10, hidden=(64,), S=50), 256, "pairs"),   # another latent size: the logits GEMM on f16 pairs
       ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(64,), S=40), 256, "evalf"), # BASELINE configs[1]'s model: evalf_rows_v
       ("vae", O.Dims(D=784, L=2, K=1, hidden=(64,), S=50), 100, "evalf"),       # configs[0]'s
       ("vae", O.Dims(D=784, L=64, K=1, hidden=(64,), S=7), 33, "evalf"),
       ("vae_gmp", O.Dims(D=784, L=32, K=10, hidden=(64,), S=40), 256, "pairs"),
       ("gmvae", O.Dims(D=784, L=32, K=10, hidden=(64,)), 512, None),            # S = 1: the chain kernels
       ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,), S=5), 64, None)]


@pytest.mark.parametrize("span", [8.0, 20.0])
@pytest.mark.parametrize("name,d,B,path", FWD, ids=[f"{n}-L{d.L}-H{d.hidden[0]}-S{d.S}-B{B}-{p}" for n, d, B, p in FWD])
def test_forward_only_saturated(name, d, B, path, span, monkeypatch):
    """gmvae_forward (scripts/runners.py:324-333 reuses run_model for the evaluation bound) at saturated parameters: every row's
    log p(x|z), log w, z and the IWAE bound against the fp64 oracle -- and the path named (the one-launch evaluation of
    csrc/evalf.hpp at the reference's default sizes; the f16-pair logits GEMM elsewhere) really ran where claimed."""
    import hip_util as H
    model = O.MODEL_NAMES[name]
    rng = np.random.default_rng(B + d.S)
    x, eps, u = O.make_inputs(d, B, model)
    p = saturate(model, d, O.init_params(model, d, rng), rng, x, span=span)
    if u is not None:
        u[0, 0], u[1, d.K - 1], u[2, :], u[3, :] = O.TINY_F32, U_MAX, U_MAX, O.TINY_F32
    flat = O.pack(model, d, p, np.float32)
    Cc = O.forward(model, d, O.unpack(model, d, flat.astype(np.float64)), x, eps, u)
    tail, rows, z, y, lg = H.hip_forward(model, d, flat, x, eps, u)
    np.testing.assert_allclose(z, Cc["z"], rtol=1e-4, atol=1e-4)
    if model == O.MODEL_GMVAE:
        np.testing.assert_allclose(y, Cc["y"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rows[:, 0], Cc["logpx"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(rows[:, 3], Cc["logw"], rtol=1e-5, atol=1e-3)
    assert abs(tail[0] / B - Cc["loss"]) <= 1e-5 * abs(Cc["loss"])
    if path:                                        # evidence that the path named ran: not the bits of the path behind it
        monkeypatch.setenv("GMVAE_NO_PLANES" if path == "pairs" else "GMVAE_NO_EVALF", "1")
        rows32 = H.hip_forward(model, d, flat, x, eps, u)[1]
        assert not np.array_equal(rows[:, 0], rows32[:, 0])
        np.testing.assert_allclose(rows[:, 0], rows32[:, 0], rtol=1e-5, atol=1e-3)


def _structured_pixels(n, D, rng):
    """MNIST-like rows: ten smooth prototypes of mostly-saturated pixels (0 / 255), 4 % of the pixels flipped, grey edges."""
    side = int(np.sqrt(D))
    f = rng.normal(size=(10, side + 6, side + 6))
    for _ in range(3):
        f = (f + np.roll(f, 1, 1) + np.roll(f, -1, 1) + np.roll(f, 1, 2) + np.roll(f, -1, 2)) / 5
    proto = np.clip((f[:, 3:-3, 3:-3] - 0.02) * 4000 + 128, 0, 255).reshape(10, -1)[:, :D]
    lab = rng.integers(0, 10, n)
    pix = proto[lab]
    flip = rng.random(pix.shape) < 0.04
    return np.where(flip, 255 - pix, pix).astype(np.uint8), lab


@pytest.mark.parametrize("model,Lz,K,H_,B,launches", [("gmvae", 64, 10, 64, 1024, M3), ("vae_gmp", 64, 10, 64, 256, M3V),
                                                       ("vae", 2, 1, 64, 100, M3V), ("gmvae", 128, 10, 512, 64, None)])
def test_step_after_300_training_steps_on_structured_pixels(model, Lz, K, H_, B, launches):
    """304 steps of the pipeline graph (binarisation inside, scripts/runners.py:44-47) at lr = 3e-3 on structured pixels, then
    ONE more step of the timed kernels at the parameters training arrived at, against the oracle at those parameters."""
    from gmvae_amd.data import DeviceDataset
    from gmvae_amd.engine import Engine
    rng = np.random.default_rng(7)
    pix, _ = _structured_pixels(8192, 784, rng)
    ds = DeviceDataset(pix, shuffle=True, seed=3)
    e = Engine(model, 784, Lz, K, [H_], random_seed=5)
    replay = e.capture_train_pipeline(ds, B, lr=3e-3, n_steps=16)
    for _ in range(19):
        replay()
    torch.cuda.synchronize()
    assert e.handoff_timeouts() == 0 and e.global_step == 304
    tl = replay.tail_log.cpu().numpy()
    assert np.isfinite(tl).all()
    flat = e.params.detach().cpu().numpy().copy()
    assert np.isfinite(flat).all()
    xb = (pix[:B].astype(np.float32) / 255.0 < rng.random((B, 784), dtype=np.float32)).astype(np.uint8)    # runners.py:45-46
    Cc, _ = graph_step_case(model, 784, Lz, K, (H_,), B, launches, params=flat, step0=304, seed=5, x=xb)
    print(f"\n[trained] {model} H={H_} B={B}: loss {tl[0, 0] / B:.1f} -> {tl[-1, 0] / B:.1f}; max |lambda| {np.abs(Cc['lam']).max():.1f}, "
          f"sigma_q in [{Cc['sig_q'].min():.2e}, {Cc['sig_q'].max():.2e}]")


@pytest.mark.parametrize("B,S,rank", [(256, 50, 0), (100, 7, 3), (37, 1, 5), (9, 100, 2), (300, 64, 0),
                                          (2100, 2, 0), (4400, 1, 1)])      # (more than 8 batch rows per workgroup: several table passes, the ring primed across them)
def test_one_launch_evaluation_draws_the_documented_noise_stream(B, S, rank):
    """csrc/evalf.hpp with its OWN Philox draws (eps = u = NULL: what bench.py --config eval_iwae and run_eval time) against the
    oracle on gmvae_noise_fill's arrays for the same (seed, step, global sample rows): a lane of the kernel asks noise_vals for
    quad 4 t + lk of its row -- the same function, the same counters, a non-zero row offset included."""
    from test_timed_path import _noise
    from gmvae_amd import _lib as L
    from gmvae_amd.engine import Engine
    d = O.Dims(D=784, L=64, K=10, hidden=(64,), S=S)
    e = Engine("gmvae", 784, 64, 10, [64], n_samples=S, random_seed=7)
    e.rank = rank
    e.global_step = 11
    x = (np.random.default_rng(B).random((B, 784)) < 0.87).astype(np.uint8)
    o = e.forward(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    assert L.step_schedule(e.dims(B), e.model) in ("mega2", "skinny", "general", "fused", "mega")      # (training's schedule: not what ran here)
    eps, u = _noise(L, B * S, 64, 10, rank * B * S, e.noise_seed, 11, True)
    flat = e.params.detach().cpu().numpy().astype(np.float64)
    Cc = O.forward(O.MODEL_GMVAE, d, O.unpack(O.MODEL_GMVAE, d, flat), x, eps, u)
    rows = o["rows"].cpu().numpy()
    np.testing.assert_allclose(o["z"].cpu().numpy(), Cc["z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o["y"].cpu().numpy(), Cc["y"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(o["logits"].cpu().numpy(), Cc["logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rows[:, 0], Cc["logpx"], rtol=1e-5)
    np.testing.assert_allclose(rows[:, 1], Cc["logq"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(rows[:, 2], Cc["logp"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(rows[:, 3], Cc["logw"], rtol=1e-5)
    tail = o["tail"].cpu().numpy().astype(np.float64)
    assert tail[4] == B and abs(tail[0] / B - Cc["loss"]) <= 1e-6 * abs(Cc["loss"])
    assert abs(tail[1] / B - Cc["nll"]) <= 1e-5 * abs(Cc["nll"]) and abs(tail[2] / B - Cc["kl"]) <= 1e-4 * max(abs(Cc["kl"]), 1.0)
    assert abs(tail[3] / B - Cc["nent"]) <= 1e-4 * max(abs(Cc["nent"]), 1.0)


def test_evaluation_reuses_its_operand_images_only_while_the_parameters_stand():
    """Engine.forward skips evalf_prep (GMVAE_SCHED_EVAL_IMAGES_VALID) when the previous pass on the same workspace left images of
    the parameters as they still are -- an evaluation walks a split on fixed parameters (scripts/runners.py:320-333).  Every writer
    must invalidate them: an in-place torch write, the eager optimizer, a train-graph replay, load_state_dict.  After each, the
    bound must follow the NEW parameters (oracle on identical noise), and between writers repeated passes must agree bit for bit."""
    from gmvae_amd.engine import Engine
    B, S = 64, 5
    d = O.Dims(D=784, L=64, K=10, hidden=(64,), S=S)
    e = Engine("gmvae", 784, 64, 10, [64], n_samples=S, random_seed=3)
    x, eps, u = O.make_inputs(d, B)
    xt, et, ut = torch.from_numpy(x).cuda(), torch.from_numpy(eps).cuda(), torch.from_numpy(u).cuda()

    def check(tag):
        a1 = e.forward(xt, et, ut)["tail"].cpu().numpy().astype(np.float64)
        a2 = e.forward(xt, et, ut)["tail"].cpu().numpy().astype(np.float64)       # (this one reuses the images)
        assert e._eval_imgs[(B, S)][:3] == e._params_state()
        flat = e.params.detach().cpu().numpy().astype(np.float64)
        ref = O.forward(O.MODEL_GMVAE, d, O.unpack(O.MODEL_GMVAE, d, flat), x, eps, u)["loss"]
        assert np.array_equal(a1, a2), tag
        assert abs(a1[0] / B - ref) <= 1e-6 * abs(ref), (tag, a1[0] / B, ref)
        return a1[0] / B

    v0 = check("start")
    with torch.no_grad():
        e.params.mul_(1.05)                              # torch-side in-place write
    v1 = check("torch write")
    e.train_step(xt[:, :], lr=1e-2)                      # eager HIP optimizer (S = 5 training step)
    v2 = check("eager step")
    sx, replay = e.capture_train_step(B, lr=1e-2, n_steps=2)
    sx.copy_(xt.unsqueeze(0).expand(2, -1, -1))
    replay()
    v3 = check("train graph")
    sd = e.state_dict()
    sd["decoder_fcnet/linear_1/b"] = sd["decoder_fcnet/linear_1/b"] + 0.25
    e.load_state_dict(sd)
    v4 = check("load_state_dict")
    assert len({v0, v1, v2, v3, v4}) == 5


def _evalf_random_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        S = int(rng.choice([1, 2, 3, 5, 9, 16, 17, 33, 50, 64, 65, 130]))
        B = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 40, 255, 256, 257, 300, 513, 1024])) if S <= 17 else int(rng.choice([1, 3, 8, 31, 64, 100, 256]))
        out.append((B, S, float(rng.choice([1.0, 0.5, 2.0])), float(rng.choice([0.0, -0.4, 1.3])), bool(rng.integers(0, 2)), int(rng.integers(0, 4))))
    return out


EVALF_RANDOM = _evalf_random_cases(24, 20261005)
EVALF_RANDOM_V = [(m,) + c for m, c in zip(["vae2", "vae64", "vae_gmp"] * 6, _evalf_random_cases(18, 20261006))]


@pytest.mark.parametrize("B,S,temp,gbias,philox,rank", EVALF_RANDOM, ids=[f"B{c[0]}-S{c[1]}-T{c[2]}-g{c[3]}-{'philox' if c[4] else 'ext'}-r{c[5]}" for c in EVALF_RANDOM])
def test_one_launch_evaluation_equals_the_general_schedule_on_random_shapes(B, S, temp, gbias, philox, rank, monkeypatch):
    """csrc/evalf.hpp against the general / chain schedules (GMVAE_NO_EVALF=1: kernels that meet the oracle elsewhere) on random
    batch sizes and sample counts -- fewer batch rows than workgroups, panels that straddle batch rows, ragged last panels, more
    than 64 samples, one row -- with its own Philox draws or external noise, a row offset, other temperatures and bias_init:
    every output (row terms, z, y, logits, the four sums) to fp32 rounding of two different summation orders."""
    from gmvae_amd.engine import Engine
    e = Engine("gmvae", 784, 64, 10, [64], n_samples=S, temperature=temp, gen_bias_init=gbias, random_seed=B * 7 + S)
    e.rank = rank
    e.global_step = 3
    rng = np.random.default_rng(B * 131 + S)
    x = torch.from_numpy((rng.random((B, 784)) < 0.87).astype(np.uint8)).cuda()
    eps = u = None
    if not philox:
        eps = torch.from_numpy(rng.standard_normal((B * S, 64)).astype(np.float32))
        u = torch.from_numpy(np.clip(rng.random((B * S, 10)).astype(np.float32), O.TINY_F32, U_MAX))
    a_ = {k: (v.cpu().numpy() if v is not None else None) for k, v in e.forward(x, eps, u).items()}
    monkeypatch.setenv("GMVAE_NO_EVALF", "1")
    b_ = {k: (v.cpu().numpy() if v is not None else None) for k, v in e.forward(x, eps, u).items()}
    assert not np.array_equal(a_["rows"], b_["rows"]) or B * S < 4          # (two different kernels ran)
    np.testing.assert_allclose(a_["z"], b_["z"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(a_["y"], b_["y"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(a_["logits"], b_["logits"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(a_["rows"], b_["rows"], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(a_["tail"][:5], b_["tail"][:5], rtol=2e-6, atol=1e-3)


@pytest.mark.parametrize("mname,B,S,temp,gbias,philox,rank", EVALF_RANDOM_V,
                         ids=[f"{c[0]}-B{c[1]}-S{c[2]}-g{c[4]}-{'philox' if c[5] else 'ext'}-r{c[6]}" for c in EVALF_RANDOM_V])
def test_one_launch_evaluation_of_the_vae_family_equals_the_general_schedule(mname, B, S, temp, gbias, philox, rank, monkeypatch):
    """evalf_rows_v (csrc/evalf.hpp: the VAE at latent 2 and 64, VAE_GMP at latent 64 / K = 10 -- BASELINE configs[0] / [1]'s
    models) against the general schedule on random batch sizes and sample counts, and against the fp64 oracle on the noise drawn."""
    from test_timed_path import _noise
    from gmvae_amd import _lib as L
    from gmvae_amd.engine import Engine
    model, Lz, K = {"vae2": ("vae", 2, 1), "vae64": ("vae", 64, 1), "vae_gmp": ("vae_gmp", 64, 10)}[mname]
    mid = O.MODEL_NAMES[model]
    e = Engine(model, 784, Lz, K, [64], n_samples=S, gen_bias_init=gbias, random_seed=B * 7 + S)
    e.rank = rank
    e.global_step = 3
    rng = np.random.default_rng(B * 131 + S)
    xn = (rng.random((B, 784)) < 0.87).astype(np.uint8)
    x = torch.from_numpy(xn).cuda()
    eps = None
    if not philox:
        eps = torch.from_numpy(rng.standard_normal((B * S, Lz)).astype(np.float32))
    a_ = {k: (v.cpu().numpy() if v is not None else None) for k, v in e.forward(x, eps).items()}
    en = eps.numpy() if eps is not None else _noise(L, B * S, Lz, K, rank * B * S, e.noise_seed, 3, False)[0]
    d = O.Dims(D=784, L=Lz, K=K, hidden=(64,), S=S, gen_bias_init=gbias)
    Cc = O.forward(mid, d, O.unpack(mid, d, e.params.detach().cpu().numpy().astype(np.float64)), xn, en, None)
    np.testing.assert_allclose(a_["z"], Cc["z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(a_["rows"][:, 0], Cc["logpx"], rtol=1e-5)
    np.testing.assert_allclose(a_["rows"][:, 1], Cc["logq"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(a_["rows"][:, 2], Cc["logp"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(a_["rows"][:, 3], Cc["logw"], rtol=1e-5)
    assert abs(a_["tail"][0] / B - Cc["loss"]) <= 1e-6 * abs(Cc["loss"])
    monkeypatch.setenv("GMVAE_NO_EVALF", "1")
    b_ = {k: (v.cpu().numpy() if v is not None else None) for k, v in e.forward(x, eps).items()}
    assert not np.array_equal(a_["rows"], b_["rows"]) or B * S < 4
    np.testing.assert_allclose(a_["rows"], b_["rows"], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(a_["tail"][:5], b_["tail"][:5], rtol=2e-6, atol=1e-3)
