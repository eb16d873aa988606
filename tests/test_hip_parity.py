"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle.

Tolerances: ELBO (loss) <= 1e-4 relative (BASELINE.json north_star); per-term
(nll / kl / nent) and per-tensor gradients are gated separately so cancellation
inside the 1e-4 * |ELBO| budget cannot hide a bug."""
import ctypes as C
import glob
import math
import os

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import hip_util
    return hip_util


def _L():
    from gmvae_amd import _lib
    return _lib


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,u8", [(64, 64, 64, False), (100, 50, 70, False), (33, 10, 784, True),
                                       (256, 128, 784, True), (130, 131, 33, False), (7, 3, 5, False)])
def test_gemm_nn(H, cfg, M, N, K, u8):
    L = _L()
    rng = np.random.default_rng(M * 1000 + N)
    A = (rng.random((M, K)) < 0.5).astype(np.uint8) if u8 else rng.normal(size=(M, K)).astype(np.float32)
    W = rng.normal(size=(K, N)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    Ad, Wd, bd = H.dev(A), H.dev(W), H.dev(b)
    Cd = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), int(u8), L.ptr(Wd), L.ptr(bd), L.ptr(Cd), M, N, K, 0, 1, cfg, 1,
                                  L.current_stream()), "gemm_test")
    ref = np.maximum(A.astype(np.float64) @ W.astype(np.float64) + b, 0)
    np.testing.assert_allclose(Cd.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * math.sqrt(K))


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (100, 70, 50), (257, 10, 128), (31, 784, 64)])
def test_gemm_nt(H, cfg, M, N, K):
    L = _L()
    rng = np.random.default_rng(M + N)
    A = rng.normal(size=(M, K)).astype(np.float32)
    W = rng.normal(size=(N, K)).astype(np.float32)
    Ad, Wd = H.dev(A), H.dev(W)
    Cd = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), 0, L.ptr(Wd), None, L.ptr(Cd), M, N, K, 1, 0, cfg, 1,
                                  L.current_stream()), "gemm_test")
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    np.testing.assert_allclose(Cd.cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * math.sqrt(K))


@pytest.mark.parametrize("form,M,N,K", [("nn", 4096, 512, 64), ("nn", 2050, 64, 64), ("nn", 37, 128, 64), ("nt_mask", 4096, 512, 128),
                                         ("nt_mask", 2061, 96, 128), ("nt", 519, 64, 128), ("nt_k8", 4096, 64, 512), ("nt_k8_add", 2057, 64, 512),
                                         ("nt_k8", 5, 64, 512), ("nt_k8", 4096, 64, 640), ("nt_k8_add", 2051, 64, 640)])
def test_rows_weight_stationary(H, form, M, N, K):
    """The row-panel layers with the weight stationary (csrc/rowsws.hpp; `cfg` 8 of gmvae_gemm_test) against fp64: exact bf16 piece
    products, so the distance is fp32 accumulation's.  Ragged row counts (the last tile's rows are clamped loads, predicated
    stores), more waves than units (M = 37 / 5), the ReLU mask incl. zeros and negative zeros, the accumulate form."""
    L = _L()
    rng = np.random.default_rng(M * 7 + N + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    A[rng.random((M, K)) < 0.1] = 0.0
    if form == "nn":
        W = rng.normal(size=(K, N)).astype(np.float32)
        side = rng.normal(size=N).astype(np.float32)
        ref = np.maximum(A.astype(np.float64) @ W.astype(np.float64) + side, 0)
        trans, relu = 0, 1
    else:
        W = rng.normal(size=(N, K)).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        trans, relu, side = 1, 0, None
        if form == "nt_mask":
            side = rng.normal(size=(M, N)).astype(np.float32)
            side[rng.random((M, N)) < 0.2] = 0.0
            side[rng.random((M, N)) < 0.05] = -0.0
            ref = np.where(side > 0, ref, 0.0)
        elif form == "nt_k8_add":
            side = rng.normal(size=(M, N)).astype(np.float32)
            ref = ref + side
    Ad, Wd = H.dev(A), H.dev(W)
    Sd = H.dev(side) if side is not None else None
    Cd = torch.full((M + 3, N), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), 0, L.ptr(Wd), L.ptr(Sd) if Sd is not None else None, L.ptr(Cd), M, N, K, trans, relu, 8, 1,
                                  L.current_stream()), "gemm_test")
    out = Cd.cpu().numpy()
    assert np.isnan(out[M:]).all()                     # nothing stored past the last row
    np.testing.assert_allclose(out[:M], ref, rtol=2e-6, atol=2e-6 * math.sqrt(K))
    # shapes outside the forms are refused, not mis-run (K = 640: the two-segment form, 512 + 128 contraction steps from one matrix)
    assert L.lib.gmvae_gemm_test(L.ptr(Ad), 0, L.ptr(Wd), None, L.ptr(Cd), M, N, K + 32, trans, relu, 8, 1, L.current_stream()) != 0


@pytest.mark.parametrize("cfg", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,u8,ns", [(64, 64, 256, False, 1), (784, 64, 300, True, 4), (10, 128, 1000, False, 3),
                                          (65, 33, 70, False, 16)])
def test_gemm_tn_splitk_bias_row(H, cfg, M, N, K, u8, ns):
    L = _L()
    rng = np.random.default_rng(K)
    A = (rng.random((K, M)) < 0.5).astype(np.uint8) if u8 else rng.normal(size=(K, M)).astype(np.float32)
    dY = rng.normal(size=(K, N)).astype(np.float32)
    Ad, Yd = H.dev(A), H.dev(dY)
    Cd = torch.full((ns, M + 1, N), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), int(u8), L.ptr(Yd), L.ptr(Yd), L.ptr(Cd), M, N, K, 2, 0, cfg, ns,
                                  L.current_stream()), "gemm_test")
    got = Cd.cpu().numpy().astype(np.float64).sum(axis=0)
    ref = np.concatenate([A.astype(np.float64).T @ dY, dY.astype(np.float64).sum(0, keepdims=True)], 0)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=3e-5 * math.sqrt(K))


@pytest.mark.parametrize("trans,M,N,K,ns", [(0, 256, 384, 160, 1), (1, 384, 128, 96, 1), (2, 128, 256, 1024, 2),
                                            (2, 256, 128, 320, 3), (0, 128, 128, 32, 1),
                                            # >= 512 tiles: the XCD-aware tile orders (all tn of a tm / all tm of a (split, tn)
                                            # consecutive on one XCD), group counts that are no multiple of 8
                                            (1, 128 * 131, 512, 64, 1), (2, 512, 128 * 43, 96, 3), (0, 128 * 67, 128 * 16, 64, 1)])
def test_gemm_big_rounds_instance(H, monkeypatch, trans, M, N, K, ns):
    """Launches whose tiles are all interior 128x128 tiles of fp32 operands with whole 32-deep rounds run the big-round
    instance (gemm.hpp big_rounds: swizzled [k/4][mn][4] LDS images, 16-byte fragment reads): every operand orientation
    (NN / NT / TN with split-K and the bias-gradient column sums) against fp64 AND against the general loop."""
    L = _L()
    rng = np.random.default_rng(M + 3 * N + 7 * K + trans)
    if trans == 0:
        A, W = rng.normal(size=(M, K)).astype(np.float32), rng.normal(size=(K, N)).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64)
        shape = (M, N)
    elif trans == 1:
        A, W = rng.normal(size=(M, K)).astype(np.float32), rng.normal(size=(N, K)).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        shape = (M, N)
    else:
        A, W = rng.normal(size=(K, M)).astype(np.float32), rng.normal(size=(K, N)).astype(np.float32)
        ref = np.concatenate([A.astype(np.float64).T @ W, W.astype(np.float64).sum(0, keepdims=True)], 0)
        shape = (ns, M + 1, N)
    Ad, Wd = H.dev(A), H.dev(W)
    outs = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("GMVAE_NO_BIG", "1")
        Cd = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
        L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), 0, L.ptr(Wd), L.ptr(Wd) if trans == 2 else None, L.ptr(Cd), M, N, K, trans, 0,
                                      2, ns, L.current_stream()), "gemm_test")
        got = Cd.cpu().numpy().astype(np.float64)
        outs.append(got.sum(axis=0) if trans == 2 else got)
    np.testing.assert_allclose(outs[0], ref, rtol=2e-5, atol=3e-5 * math.sqrt(K))
    np.testing.assert_allclose(outs[0], outs[1], rtol=2e-5, atol=3e-5 * math.sqrt(K))


@pytest.mark.parametrize("trans,M,N,K,ns", [(0, 256, 384, 160, 1), (1, 384, 128, 96, 1), (2, 128, 256, 1024, 2),
                                            (2, 256, 128, 320, 3), (0, 128, 128, 32, 1), (1, 128 * 131, 512, 64, 1),
                                            (2, 512, 128 * 43, 96, 3), (0, 128 * 67, 128 * 16, 64, 1),
                                            (0, 512, 256, 64, 1), (1, 768, 128, 2048, 1), (2, 256, 256, 4096, 4)])
def test_gemm_plane_rounds_instance(H, monkeypatch, trans, M, N, K, ns):
    """Operands split ONCE into planes of 16-bit pieces (blocked by 16) and multiplied as piece products with fp32 accumulation
    through the LDS-DMA rings of gemm.hpp: f16 PAIRS under per-tensor power-of-two scales, three piece products per product
    (plane_rounds2h -- 32-k steps of v_mfma_f32_16x16x32_f16 over two images: the K values here give 1, 2, 3, 4, 5, 16, 32
    and 64 steps per tile, on both ring depths -- `cfg` 6 of gmvae_gemm_test: amax_abs + amax_final + split_pairs_b16 first) and bf16 TRIPLES, six exact
    piece products (plane_rounds3, `cfg` 4) -- every operand orientation (NN / NT / TN with split-K and the bias-gradient
    column sums) to fp32-GEMM accuracy against fp64, and no further from it than the fp32 MFMA instance is."""
    L = _L()
    err = []
    rng = np.random.default_rng(M + 3 * N + 7 * K + trans)
    if trans == 0:
        A, W = rng.normal(size=(M, K)).astype(np.float32), rng.normal(size=(K, N)).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64)
        shape = (M, N)
    elif trans == 1:
        A, W = rng.normal(size=(M, K)).astype(np.float32), rng.normal(size=(N, K)).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T
        shape = (M, N)
    else:
        A, W = rng.normal(size=(K, M)).astype(np.float32), rng.normal(size=(K, N)).astype(np.float32)
        ref = np.concatenate([A.astype(np.float64).T @ W, W.astype(np.float64).sum(0, keepdims=True)], 0)
        shape = (ns, M + 1, N)
    Ad, Wd = H.dev(A), H.dev(W)
    for cfg in (6, 6, 4, 2):
        # (the pairs once per workgroup tile: 128 x 128, and -- where M is a multiple of 256 -- the 8-wave 256 x 128 instance)
        monkeypatch.setenv("GMVAE_PAIRS_BM", "128" if len(err) == 0 else "256")
        Cd = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
        L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), 0, L.ptr(Wd), L.ptr(Wd) if trans == 2 else None, L.ptr(Cd), M, N, K, trans, 0,
                                      cfg, ns, L.current_stream()), "gemm_test")
        got = Cd.cpu().numpy().astype(np.float64)
        got = got.sum(axis=0) if trans == 2 else got
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=3e-5 * math.sqrt(K))
        err.append(np.abs(got - ref).max())
    assert max(err[:3]) <= 2.0 * err[3] + 1e-6, err


@pytest.mark.parametrize("trans", [0, 1, 2])
@pytest.mark.parametrize("case", ["huge x tiny", "20 binades inside a tensor", "sparse rows", "one operand all zero", "a NaN", "subnormal maximum"])
def test_gemm_pair_scales(H, trans, case):
    """The f16 pairs' scaling (gemm.hpp pair_scale / split_pairs_b16): the power-of-two scale comes from the tensor's own
    largest magnitude, so tensors of ANY magnitude multiply to fp32-GEMM accuracy (fp16's range never shows); inside one tensor
    an element keeps its 22 bits down to 2^-29 of the largest (the shifted second piece has the first one's exponent) -- rows 10^6
    times smaller than others still come out to 1e-5 of THEIR OWN scale; an all-zero operand gives exact zeros; a NaN
    propagates to the outputs it touches and to no others."""
    L = _L()
    M, N, K, ns = 256, 128, 512, 2
    rng = np.random.default_rng(trans + 17)
    a_shape = (K, M) if trans == 2 else (M, K)
    w_shape = (N, K) if trans == 1 else (K, N)
    A, W = rng.normal(size=a_shape), rng.normal(size=w_shape)
    row_scale = np.ones(M)
    if case == "huge x tiny":
        A *= 3e30; W *= 2e-34
    elif case == "20 binades inside a tensor":
        row_scale = 2.0 ** -rng.integers(0, 21, size=M)
    elif case == "sparse rows":
        row_scale = np.where(rng.random(M) < 0.5, 1.0, 1e-6)
    elif case == "one operand all zero":
        W[:] = 0.0
    elif case == "subnormal maximum":
        A *= 1e-41 / np.abs(A).max()
    A = A * (row_scale[None, :] if trans == 2 else row_scale[:, None])
    A, W = A.astype(np.float32), W.astype(np.float32)
    if case == "a NaN":
        if trans == 2: A[5, 7] = np.nan
        else: A[7, 5] = np.nan
    A64, W64 = A.astype(np.float64), W.astype(np.float64)
    ref = A64 @ W64 if trans == 0 else (A64 @ W64.T if trans == 1 else A64.T @ W64)
    shape = (ns, M + 1, N) if trans == 2 else (M, N)
    Cd = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
    Ad, Wd = H.dev(A), H.dev(W)
    L.check(L.lib.gmvae_gemm_test(L.ptr(Ad), 0, L.ptr(Wd), None, L.ptr(Cd), M, N, K, trans, 0, 6, ns, L.current_stream()), "gemm_test")
    got = Cd.cpu().numpy().astype(np.float64)
    got = got.sum(axis=0)[:M] if trans == 2 else got
    if case == "a NaN":
        assert np.isnan(got[7]).all() and not np.isnan(np.delete(got, 7, axis=0)).any()
        got, ref = np.delete(got, 7, axis=0), np.delete(ref, 7, axis=0)
    if case == "one operand all zero":
        assert (got == 0).all()
        return
    # every output ROW against the scale of that row's own products (not the tensor's): sqrt(K) |a_row| |w| typical
    row_mag = np.sqrt((A64 ** 2).sum(0 if trans == 2 else 1))
    if case == "a NaN": row_mag = np.delete(row_mag, 7)
    w_mag = np.sqrt((W64 ** 2).mean())
    err = np.abs(got - ref).max(axis=1) / (row_mag * w_mag)
    assert err.max() <= (1e-3 if case == "subnormal maximum" else 1e-5), (case, err.max())       # (fp32 denormal INPUTS carry few bits themselves)


# ------------------------------------------------------------ full step
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(f)[:-4] for f in GOLDEN])
def test_step_matches_golden(H, path):
    z = np.load(path)
    model = int(z["model"])
    d = O.Dims(D=int(z["D"]), L=int(z["L"]), K=int(z["K"]), S=int(z["S"]), hidden=tuple(int(h) for h in z["hidden"]))
    B = z["x"].shape[0]
    u = z["u"] if "u" in z.files else None
    gs, tail = H.hip_step(model, d, z["params"], z["x"], z["eps"], u)
    assert abs(tail[0] / B - z["loss"]) <= 1e-4 * abs(z["loss"])
    assert abs(tail[1] / B - z["nll"]) <= 1e-4 * abs(z["nll"])
    assert abs(tail[2] / B - z["kl"]) <= 1e-4 * max(abs(float(z["kl"])), 1.0)
    assert abs(tail[3] / B - z["nent"]) <= 1e-4 * max(abs(float(z["nent"])), 1.0)
    lay, P, _ = O.param_layout(model, d)
    for name, shape, off in lay:
        n = int(np.prod(shape))
        ref = z["grads"][off:off + n].astype(np.float64)
        got = gs[off:off + n] / B
        assert np.abs(got - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6), name


STEP_CASES = [
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,)), 256),
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,)), 1024),
    ("gmvae", O.Dims(D=784, L=8, K=10, hidden=(64,)), 16),                 # run_gmvae.py defaults
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,)), 64),              # bin/run_train.sh
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,)), 96),                # last partial batch of B=256
    ("gmvae", O.Dims(D=784, L=16, K=10, hidden=(48, 32)), 50),
    ("gmvae", O.Dims(D=300, L=16, K=64, hidden=(40,), S=5), 24),
    ("gmvae", O.Dims(D=97, L=5, K=3, hidden=(), temperature=0.5), 9),
    ("gmvae", O.Dims(D=64, L=8, K=4, hidden=(16,), sigma_min=0.9, raw_sigma_bias=0.25), 32),   # clamp active
    ("vae", O.Dims(D=784, L=2, K=1, hidden=(64,)), 100),                   # BASELINE config 1
    ("vae", O.Dims(D=784, L=8, K=1, hidden=(32,), S=4), 20),
    ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(64,)), 256),             # BASELINE config 2
    ("vae_gmp", O.Dims(D=200, L=24, K=64, hidden=(32,), S=3), 17),
    ("vae_gmp", O.Dims(D=128, L=7, K=5, hidden=(20, 20), gen_bias_init=0.2), 33),
    # single-launch (mega) schedule of the VAE family: std-normal prior, learned mixture prior (K > 32: two
    # components per lane), ragged last panel, sigma clamp and decoder bias offset
    ("vae", O.Dims(D=784, L=64, K=1, hidden=(64,)), 1024),
    ("vae", O.Dims(D=784, L=16, K=1, hidden=(32,), sigma_min=0.8, gen_bias_init=-0.3), 37),
    ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(64,)), 1000),
    ("vae_gmp", O.Dims(D=208, L=24, K=40, hidden=(48,)), 50),
    # latent sizes that are even but no multiple of 8 (padded head tiles, zero k rows in z * Wd0, ragged eps rows)
    ("gmvae", O.Dims(D=784, L=6, K=10, hidden=(64,)), 41),
    ("vae_gmp", O.Dims(D=400, L=10, K=7, hidden=(32,)), 23),
    ("vae", O.Dims(D=784, L=2, K=1, hidden=(64,)), 1024),
    # the general schedule at thousands of rows WITHOUT the plane GEMMs: the first layers over the uint8 batch as bf16 piece
    # products (first_layers_u8bf) and the forward row-panel layers -- the y part + prior net side by side, encoder_gmm's and the
    # decoder's hidden layers -- as register-direct piece products (rows_nn_bf6: R >= 2048, K % 32 = 0, widths % 64 = 0); ragged
    # last row tiles; IWAE rows (R = B S) in the second case
    ("gmvae", O.Dims(D=256, L=32, K=32, hidden=(128, 64)), 2090),
    ("gmvae", O.Dims(D=208, L=64, K=64, hidden=(64, 128), S=3), 700),
    ("vae", O.Dims(D=256, L=32, K=1, hidden=(128, 128)), 2100),
]
# hidden_activation_fn other than ReLU (scripts/base.py:19,90,153; gmvae.py:282, vae.py:196): tanh / sigmoid / ELU in the grouped
# GEMM's epilogues (forward: the activation; data gradient: its derivative from the kept activation) -- all three models, two hidden
# layers, IWAE rows, the reference's default sizes (which must leave the ReLU-only one-launch schedules), thousands of rows
ACT_CASES = [
    ("gmvae", O.Dims(D=300, L=16, K=10, hidden=(48, 32), act="tanh"), 50),
    ("vae", O.Dims(D=784, L=8, K=1, hidden=(64,), S=3, act="sigmoid"), 40),
    ("vae_gmp", O.Dims(D=200, L=24, K=12, hidden=(32,), act="elu"), 33),
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,), act="elu"), 256),
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,), act="tanh"), 64),
    ("gmvae", O.Dims(D=256, L=32, K=32, hidden=(128, 64), act="sigmoid"), 2090),
]


@pytest.mark.parametrize("name,d,B", ACT_CASES, ids=[f"{n}-{d.act}-D{d.D}-L{d.L}-H{'x'.join(map(str, d.hidden))}-S{d.S}-B{B}" for n, d, B in ACT_CASES])
def test_step_with_other_hidden_activations_matches_oracle(H, name, d, B):
    model = O.MODEL_NAMES[name]
    assert _L().step_schedule(H.dims_of(d, B), model) == "general"
    rng = np.random.default_rng(B)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)
    # forward-only evaluation and one conditional's MLP on the same activation
    Cc = O.forward(model, d, O.unpack(model, d, O.pack(model, d, p, np.float32).astype(np.float64)), x, eps, u, np.float64)
    tail, rows = H.hip_forward(model, d, O.pack(model, d, p, np.float32), x, eps, u)[:2]
    assert abs(tail[0] / B - Cc["loss"]) <= 1e-4 * abs(Cc["loss"])


def test_large_row_cases_take_the_general_schedule(H):
    """The three STEP_CASES meant for the general schedule's row-panel kernels really take that schedule."""
    for name, d, B in STEP_CASES[-3:]:
        assert _L().step_schedule(H.dims_of(d, B), O.MODEL_NAMES[name]) == "general", (name, d, B)
        assert B * d.S >= 2048


@pytest.mark.parametrize("name,d,B", STEP_CASES, ids=[f"{n}-D{d.D}-L{d.L}-K{d.K}-H{'x'.join(map(str, d.hidden))}-S{d.S}-B{B}"
                                                       for n, d, B in STEP_CASES])
def test_step_matches_oracle(H, name, d, B):
    model = O.MODEL_NAMES[name]
    rng = np.random.default_rng(B)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)


def _random_cases(n, seed):
    """Seeded random model shapes (widths that are and are not multiples of the 32 / 64 / 128 tile sizes, 0-3 hidden
    layers, IWAE samples, batches from one ragged panel to a few hundred rows): the tile-configuration choice, the
    big-round eligibility, the slab-count ranges and the schedules' size gates all depend on them."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        model = ["gmvae", "vae", "vae_gmp"][i % 3]
        nl = int(rng.integers(0, 4))
        hidden = tuple(int(rng.choice([16, 24, 40, 64, 96, 128, 200])) for _ in range(nl))
        d = O.Dims(D=int(rng.choice([64, 100, 128, 256, 300, 784])), L=int(rng.choice([2, 5, 8, 16, 30, 64])),
                   K=1 if model == "vae" else int(rng.choice([2, 7, 10, 16, 33])), hidden=hidden,
                   S=int(rng.choice([1, 1, 2, 3])), temperature=float(rng.choice([1.0, 0.7])))
        out.append((model, d, int(rng.choice([1, 7, 16, 50, 128, 257]))))
    return out


RANDOM_CASES = _random_cases(18, 20261004)


@pytest.mark.parametrize("name,d,B", RANDOM_CASES, ids=[f"{n}-D{d.D}-L{d.L}-K{d.K}-H{'x'.join(map(str, d.hidden)) or 'none'}-S{d.S}-B{B}"
                                                         for n, d, B in RANDOM_CASES])
def test_step_matches_oracle_on_random_shapes(H, name, d, B):
    model = O.MODEL_NAMES[name]
    rng = np.random.default_rng(B * 131 + d.D)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)



def test_config5_shapes_iwae(H):
    """BASELINE config 5 shapes (D=3072, K=64, S=50, H=512) at a batch the oracle finishes in seconds.
    ELBO and every gradient tensor at 1e-4.  log w ~ -2100 has an fp32 ulp of 2.4e-4 and the IWAE weights
    exp(log w - lse) inherit the ABSOLUTE error of log w, so the device keeps log w in fp64 from the per-tile partial sums
    on (kernels.hpp row_terms / iwae_rows) and forms the weights from differences to the group's maximum."""
    d = O.Dims(D=3072, L=64, K=64, hidden=(512,), S=50)
    p = O.init_params(O.MODEL_GMVAE, d, np.random.default_rng(0))
    x, eps, u = O.make_inputs(d, 8)
    H.compare_step(O.MODEL_GMVAE, d, p, x, eps, u)


@pytest.mark.parametrize("model", [O.MODEL_GMVAE, O.MODEL_VAE])
def test_big_round_gemm_inside_the_step_iwae(H, monkeypatch, model):
    """The big-round GEMM instance inside a step: sizes at which the decoder launches are made of interior 128 x 128
    tiles (R = B*S = 128 rows, H = 128, D = 256) with the 128 x 128 configuration forced -- forward + Bernoulli
    epilogue, data gradient with the IWAE row scale, weight gradient with the per-k IWAE scale and the bias-gradient
    column sums -- against the oracle, and NOT bit-identical to the general loop (evidence that the instance ran)."""
    monkeypatch.setenv("GMVAE_FORCE_CFG", "2")
    d = O.Dims(D=256, L=64, K=10, hidden=(128,), S=4)
    rng = np.random.default_rng(5)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, 32, model)
    H.compare_step(model, d, p, x, eps, u)
    flat = O.pack(model, d, p, np.float32)
    g_big, _ = H.hip_step(model, d, flat, x, eps, u)
    monkeypatch.setenv("GMVAE_NO_BIG", "1")
    g_gen, _ = H.hip_step(model, d, flat, x, eps, u)
    assert not np.array_equal(g_big, g_gen)
    np.testing.assert_allclose(g_big, g_gen, rtol=0, atol=2e-4 * np.abs(g_gen).max())


@pytest.mark.parametrize("exact", [0, 1], ids=["f16 pairs", "bf16 triples"])
@pytest.mark.parametrize("model,S,hidden", [(O.MODEL_GMVAE, 4, (128,)), (O.MODEL_VAE, 1, (64, 128)), (O.MODEL_VAE_GMP, 2, (128,))])
def test_plane_gemms_inside_the_step(H, monkeypatch, model, S, hidden, exact):
    """The top decoder layer's three GEMMs on pre-split operands (gemm.hpp plane_rounds2 / plane_rounds3; by default from 4096
    rows, forced here at R = B*S = 128 / 256 rows, H = 128, D = 256), in both piece forms -- f16 pairs (the default: the
    activation's largest magnitude from the producing launch or amax_abs, the weight's from amax_abs, amax_final,
    split_pairs_b16; (sigmoid - x) written as pairs under a fixed scale by the Bernoulli epilogue) and bf16 triples
    (GMVAE_PLANES_EXACT=1: split_planes_b16, three planes) -- the IWAE row weights riding on the activation's pieces and
    weighing the bias gradient's column sums -- against the oracle at the step's gates, NOT bit-identical to the fp32 MFMA
    instance (evidence that the path ran) and as close to it as fp32 rounding."""
    monkeypatch.setenv("GMVAE_PLANES_MINROWS", "128")
    monkeypatch.setenv("GMVAE_PLANES_EXACT", str(exact))
    d = O.Dims(D=256, L=64, K=10, hidden=hidden, S=S)
    rng = np.random.default_rng(6)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    B = 256 // S if S > 1 else 128
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)
    flat = O.pack(model, d, p, np.float32)
    g_pl, _ = H.hip_step(model, d, flat, x, eps, u)
    monkeypatch.setenv("GMVAE_PLANES_EXACT", str(1 - exact))
    g_other, _ = H.hip_step(model, d, flat, x, eps, u)
    assert not np.array_equal(g_pl, g_other)                     # (the two piece forms are different arithmetic)
    monkeypatch.setenv("GMVAE_NO_PLANES", "1")
    g_f32, _ = H.hip_step(model, d, flat, x, eps, u)
    assert not np.array_equal(g_pl, g_f32)
    np.testing.assert_allclose(g_pl, g_f32, rtol=0, atol=2e-5 * np.abs(g_f32).max())


def test_plane_gemms_at_config5_dims_under_natural_gating(H, monkeypatch):
    """BASELINE configs[4]'s dimensions (D = 3072, K = 64, H = 512, S = 50) with NO GMVAE_PLANES_* switch set, at the smallest
    batch that passes every gate of planes_ok by itself: R = B * S must be >= 4096 AND a multiple of 128 (whole 128-row
    tiles), i.e. B a multiple of 64 -- B = 128, R = 6,400 sample rows.  The step then runs "general+planes" exactly as the
    25,600-row shard of the benchmark does -- logits + Bernoulli epilogue, the data gradient and the 6-of-9-piece weight
    gradient over a 6,400-long contraction in 5 split-K slabs (the rule of DESIGN.md 3.5), the IWAE row weights on the activation's
    pieces, the thin layers and data gradients on the weight-stationary row kernels (csrc/rowsws.hpp) -- and every
    loss term and every gradient tensor is compared with the fp64 oracle at the step's gates (NumPy needs ~20 s there)."""
    for k in ("GMVAE_PLANES_MINROWS", "GMVAE_NO_PLANES", "GMVAE_NSPLIT_SMALL", "GMVAE_FORCE_CFG", "GMVAE_NO_BIG"):
        monkeypatch.delenv(k, raising=False)
    d = O.Dims(D=3072, L=64, K=64, hidden=(512,), S=50)
    B = 128
    assert _L().step_schedule(H.dims_of(d, B), O.MODEL_GMVAE) == "general+planes"
    rng = np.random.default_rng(11)
    p = O.init_params(O.MODEL_GMVAE, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B)
    H.compare_step(O.MODEL_GMVAE, d, p, x, eps, u)


@pytest.mark.parametrize("B,S", [(41, 50), (2049, 1)])
def test_weight_stationary_rows_inside_the_step_with_a_ragged_last_tile(H, monkeypatch, B, S):
    """The general schedule at >= 2048 sample rows with 64 components, 64 latents and a 512-wide hidden layer takes every form of
    csrc/rowsws.hpp -- y / z -> 512 units (K = 64), dz = dhd Wd0^T (K = 512 over eight waves), dhg = dqp Wg1^T under the ReLU mask
    (K = 128), dy = dhg Wg0y^T + dpp Wp^T (two segments) -- here with R = 2,050 / 2,049 rows: the last row tile has 2 / 1 rows
    (clamped loads, predicated stores, run once behind the loops), D = 80 keeps the oracle quick.  Against the fp64 oracle at the
    step's gates, and against the same step with GMVAE_NO_RWS=1."""
    for k in ("GMVAE_NO_RWS", "GMVAE_PLANES_MINROWS", "GMVAE_NO_PLANES"):
        monkeypatch.delenv(k, raising=False)
    d = O.Dims(D=80, L=64, K=64, hidden=(512,), S=S)
    rng = np.random.default_rng(B)
    p = O.init_params(O.MODEL_GMVAE, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B)
    H.compare_step(O.MODEL_GMVAE, d, p, x, eps, u)
    flat = O.pack(O.MODEL_GMVAE, d, p, np.float32)
    g_ws, t_ws = H.hip_step(O.MODEL_GMVAE, d, flat, x, eps, u)
    monkeypatch.setenv("GMVAE_NO_RWS", "1")
    g_old, t_old = H.hip_step(O.MODEL_GMVAE, d, flat, x, eps, u)
    assert not np.array_equal(g_ws, g_old)                       # (the switch did switch)
    np.testing.assert_allclose(g_ws, g_old, rtol=0, atol=2e-5 * np.abs(g_old).max())
    np.testing.assert_allclose(t_ws[:4], t_old[:4], rtol=1e-6)


@pytest.mark.parametrize("ns", [1, 3, 16])
def test_top_weight_gradient_is_the_same_sum_in_any_number_of_slabs(H, monkeypatch, ns):
    """The plane launch's weight gradient is split over the rows into a number of slabs chosen by a rule (gmvae_hip.hip: as many
    as put its tiles on the chip once); finalize_grads sums whatever count the launch wrote.  D = 256, hidden 128, S = 32, B = 128
    (R = 4,096 rows: up to 16 slabs), planes forced: gradients with 1, 3 and 16 slabs against the rule's own count -- the same sums
    in another order, so equal to fp32 summation error -- and the losses bit-identical (the forward pass does not depend on it)."""
    monkeypatch.setenv("GMVAE_PLANES_MINROWS", "128")
    monkeypatch.setenv("GMVAE_NO_SKINNY", "1")
    monkeypatch.delenv("GMVAE_NSPLIT_TOP", raising=False)
    d = O.Dims(D=256, L=16, K=10, hidden=(128,), S=32)
    rng = np.random.default_rng(5)
    p = O.init_params(O.MODEL_GMVAE, d, rng)
    flat = O.pack(O.MODEL_GMVAE, d, p, np.float32)
    x, eps, u = O.make_inputs(d, 128)
    g0, t0 = H.hip_step(O.MODEL_GMVAE, d, flat, x, eps, u)
    monkeypatch.setenv("GMVAE_NSPLIT_TOP", str(ns))
    g1, t1 = H.hip_step(O.MODEL_GMVAE, d, flat, x, eps, u)
    assert np.isfinite(g1).all() and np.array_equal(t0[:4], t1[:4])
    np.testing.assert_allclose(g1, g0, rtol=0, atol=3e-6 * np.abs(g0).max())


def _random_plane_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        model = ["gmvae", "vae", "vae_gmp"][i % 3]
        S = int(rng.choice([1, 2, 4]))
        R = int(rng.choice([128, 256, 384]))
        hidden = tuple([int(rng.choice([64, 96]))] * int(rng.integers(0, 2)) + [int(rng.choice([128, 256]))])
        d = O.Dims(D=int(rng.choice([128, 256, 384])), L=int(rng.choice([8, 20, 64])), K=1 if model == "vae" else int(rng.choice([3, 10])),
                   hidden=hidden, S=S)
        out.append((model, d, R // S))
    return out


RANDOM_PLANE_CASES = _random_plane_cases(9, 20261006)


@pytest.mark.parametrize("name,d,B", RANDOM_PLANE_CASES, ids=[f"{n}-D{d.D}-L{d.L}-H{'x'.join(map(str, d.hidden))}-S{d.S}-B{B}" for n, d, B in RANDOM_PLANE_CASES])
def test_plane_gemms_on_random_shapes(H, monkeypatch, name, d, B):
    """The plane path (forced from 128 rows) on random eligible shapes -- one or two hidden layers (the activation's planes from
    the producing GEMM's epilogue either way), IWAE samples, all three models -- against the oracle."""
    monkeypatch.setenv("GMVAE_PLANES_MINROWS", "128")
    monkeypatch.setenv("GMVAE_NO_SKINNY", "1")
    L = _L()
    model = O.MODEL_NAMES[name]
    assert L.step_schedule(H.dims_of(d, B), model) == "general+planes"
    rng = np.random.default_rng(B + d.D)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)


def test_many_splits_for_small_weight_gradients_and_few_for_batch_row_ones(H, monkeypatch):
    """General schedule, S > 1: the per-range slab counts of finalize_grads -- weight gradients contracted over the B batch
    rows written to num_splits(B) slabs, those with few outputs and a contraction over all B*S rows to MORE slabs than the
    rest (forced here at a size the oracle covers; by default from R = 6800 rows) -- against the oracle."""
    monkeypatch.setenv("GMVAE_NSPLIT_SMALL", "12")            # R = 1200 rows: 4 slabs by default, 1 for the B = 300-row products
    d = O.Dims(D=96, L=12, K=7, hidden=(40, 24), S=4)
    rng = np.random.default_rng(9)
    for model in (O.MODEL_GMVAE, O.MODEL_VAE, O.MODEL_VAE_GMP):
        p = O.init_params(model, d, rng)
        for k in p:
            if k.endswith("/b"):
                p[k] = rng.normal(0, 0.05, p[k].shape)
        x, eps, u = O.make_inputs(d, 300, model)
        H.compare_step(model, d, p, x, eps, u)


@pytest.mark.parametrize("env", [{}, {"GMVAE_NO_MEGA": "1"}, {"GMVAE_NO_FUSED": "1"}],
                         ids=["mega", "chain-kernels", "general-schedule"])
@pytest.mark.parametrize("B,L", [(1024, 64), (40, 16), (7, 8)])
def test_all_three_schedules_match_oracle(H, env, B, L, monkeypatch):
    """The same eligible configuration through (a) the single-launch mega kernel, (b) the chain kernels,
    (c) the general one-launch-per-level schedule: each must meet the oracle tolerance on its own."""
    for k in ("GMVAE_NO_MEGA", "GMVAE_NO_FUSED"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    d = O.Dims(D=784, L=L, K=10, hidden=(64,))
    rng = np.random.default_rng(B + L)
    p = O.init_params(O.MODEL_GMVAE, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B)
    H.compare_step(O.MODEL_GMVAE, d, p, x, eps, u)


@pytest.mark.parametrize("D,K,expect", [(784, 10, 541.124804), (3072, 64, 2125.189256)])
def test_kat_zero_weights_on_gpu(H, D, K, expect):
    """all-zero parameters -> loss = D ln2 - ln K for any data/noise."""
    d = O.Dims(D=D, L=64, K=K, hidden=(64,))
    lay, P, _ = O.param_layout(O.MODEL_GMVAE, d)
    x, eps, u = O.make_inputs(d, 32)
    _, tail = H.hip_step(O.MODEL_GMVAE, d, np.zeros(P, np.float32), x, eps, u)
    assert tail[0] / 32 == pytest.approx(expect, rel=2e-6)
    assert abs(tail[2] / 32) < 1e-5
    assert tail[3] / 32 == pytest.approx(-math.log(K), rel=1e-5)


def test_forward_outputs(H):
    d = O.Dims(D=784, L=16, K=10, hidden=(64,), S=2)
    model = O.MODEL_GMVAE
    p = O.init_params(model, d, np.random.default_rng(5))
    x, eps, u = O.make_inputs(d, 40)
    flat = O.pack(model, d, p, np.float32)
    Cc = O.forward(model, d, O.unpack(model, d, flat.astype(np.float64)), x, eps, u)
    tail, rows, z, y, lg = H.hip_forward(model, d, flat, x, eps, u)
    np.testing.assert_allclose(z, Cc["z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y, Cc["y"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(lg, Cc["logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rows[:, 0], Cc["logpx"], rtol=1e-5)
    np.testing.assert_allclose(rows[:, 3], Cc["logw"], rtol=1e-5)
    assert tail[0] / 40 == pytest.approx(Cc["loss"], rel=1e-5)


@pytest.mark.parametrize("name,D,hidden,S,B", [("gmvae", 784, (64,), 4, 64), ("gmvae", 200, (32, 96), 2, 128), ("vae_gmp", 784, (64,), 8, 32),
                                               ("vae", 1000, (160,), 1, 256)])
def test_forward_only_logits_gemm_on_pairs(H, monkeypatch, name, D, hidden, S, B):
    """Forward-only evaluation (gmvae_forward, S importance samples per row) at thousands of rows runs its logits GEMM on f16
    pairs whatever D (gmvae_hip.hip fwd_pairs_ok: the weight's planes are zero-padded to whole 128-column tiles, the last
    column tile's missing quads are predicated off in the Bernoulli epilogue) -- forced here at R = B*S = 256 rows with
    GMVAE_PLANES_MINROWS: D = 784 = 6 x 128 + 16, 200 and 1000 (no multiples of the tile), hidden widths of 64 / 96 / 160; every row's
    log p(x|z), log w, z and the bound against the fp64 oracle, and NOT bit-identical to the fp32 MFMA path (evidence that the
    pairs ran) while equal to it within fp32 rounding."""
    model = O.MODEL_NAMES[name]
    d = O.Dims(D=D, L=16, K=10 if name != "vae" else 1, hidden=hidden, S=S)
    p = O.init_params(model, d, np.random.default_rng(D + S))
    x, eps, u = O.make_inputs(d, B, model)
    flat = O.pack(model, d, p, np.float32)
    Cc = O.forward(model, d, O.unpack(model, d, flat.astype(np.float64)), x, eps, u)
    monkeypatch.setenv("GMVAE_PLANES_MINROWS", "256")
    tail, rows, z = H.hip_forward(model, d, flat, x, eps, u)[:3]
    np.testing.assert_allclose(z, Cc["z"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rows[:, 0], Cc["logpx"], rtol=1e-5)
    np.testing.assert_allclose(rows[:, 3], Cc["logw"], rtol=1e-5)
    assert tail[0] / B == pytest.approx(Cc["loss"], rel=1e-5)
    monkeypatch.setenv("GMVAE_NO_PLANES", "1")
    rows_f32 = H.hip_forward(model, d, flat, x, eps, u)[1]
    assert not np.array_equal(rows[:, 0], rows_f32[:, 0])
    np.testing.assert_allclose(rows[:, 0], rows_f32[:, 0], rtol=2e-6)


def _random_forward_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        model = ["gmvae", "vae", "vae_gmp"][i % 3]
        S = int(rng.choice([1, 2, 4, 8]))
        R = int(rng.choice([128, 256, 384]))
        hidden = tuple([int(rng.choice([24, 64]))] * int(rng.integers(0, 2)) + [int(rng.choice([32, 64, 96, 160]))])
        d = O.Dims(D=int(rng.choice([8, 104, 200, 784, 904])), L=int(rng.choice([2, 8, 20, 64])),
                   K=1 if model == "vae" else int(rng.choice([1, 3, 10, 16, 17])), hidden=hidden, S=S)
        out.append((model, d, R // S))
    return out


RANDOM_FORWARD_CASES = _random_forward_cases(12, 20261007)


@pytest.mark.parametrize("name,d,B", RANDOM_FORWARD_CASES,
                         ids=[f"{n}-D{d.D}-L{d.L}-K{d.K}-H{'x'.join(map(str, d.hidden))}-S{d.S}-B{B}" for n, d, B in RANDOM_FORWARD_CASES])
def test_forward_only_on_random_shapes(H, monkeypatch, name, d, B):
    """Forward-only passes on random shapes with the forward pairs forced from 128 rows: D below, at and beyond one 128-column
    tile and no multiple of it, hidden widths of 32 - 160, mixture sizes on both sides of rows_small_k's K <= 16 (and of
    y_head_fwd's four-rows-per-wave form), IWAE samples -- every row's log p(x|z), log w and z against the fp64 oracle."""
    monkeypatch.setenv("GMVAE_PLANES_MINROWS", "128")
    model = O.MODEL_NAMES[name]
    rng = np.random.default_rng(B + d.D + d.K)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    flat = O.pack(model, d, p, np.float32)
    Cc = O.forward(model, d, O.unpack(model, d, flat.astype(np.float64)), x, eps, u)
    tail, rows, z, y, lg = H.hip_forward(model, d, flat, x, eps, u)
    np.testing.assert_allclose(z, Cc["z"], rtol=1e-4, atol=1e-5)
    if model == O.MODEL_GMVAE:
        np.testing.assert_allclose(y, Cc["y"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rows[:, 0], Cc["logpx"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rows[:, 3], Cc["logw"], rtol=1e-5, atol=1e-4)
    assert tail[0] / B == pytest.approx(Cc["loss"], rel=1e-5)


# ----------------------------------------------------------------- Adam
def test_adam_tf_three_steps(H):
    """Against the oracle run in fp32 (TF's own arithmetic type) tightly, and in fp64 loosely."""
    L = _L()
    rng = np.random.default_rng(2)
    P = 1003
    th = rng.normal(size=P).astype(np.float32)
    m = np.zeros(P, np.float32)
    v = np.zeros(P, np.float32)
    th64, m64, v64 = th.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    td, md, vd = H.dev(np.pad(th, (0, 1))), H.dev(np.pad(m, (0, 1))), H.dev(np.pad(v, (0, 1)))
    for t in range(1, 4):
        g = (rng.normal(size=P) * (1e-9 if t == 2 else 1.0)).astype(np.float32)
        gd = H.dev(np.pad(g, (0, 1)) * 8.0)
        L.check(L.lib.adam_tf_step(L.ptr(td), L.ptr(md), L.ptr(vd), L.ptr(gd), P, 1e-3, 0.9, 0.999, 1e-8, t, None,
                                   1.0 / 8.0, None, None, L.current_stream()), "adam")
        th, m, v = O.adam_tf_step(th, m, v, g, t, dtype=np.float32)
        th64, m64, v64 = O.adam_tf_step(th64, m64, v64, g.astype(np.float64), t, dtype=np.float64)
    np.testing.assert_allclose(td.cpu().numpy()[:P], th, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(vd.cpu().numpy()[:P], v, rtol=2e-6, atol=1e-30)
    np.testing.assert_allclose(md.cpu().numpy()[:P], m, rtol=2e-6, atol=5e-8)    # FMA contraction near m ~ 0
    np.testing.assert_allclose(td.cpu().numpy()[:P], th64, rtol=1e-4, atol=1e-6)


def test_adam_device_counter_and_scale(H):
    L = _L()
    P = 64
    f32 = dict(dtype=torch.float32, device="cuda")
    td, md, vd = torch.ones(P, **f32), torch.zeros(P, **f32), torch.zeros(P, **f32)
    gd = torch.full((P,), 6.0, **f32)
    tdev = torch.tensor([3, 0], dtype=torch.int64, device="cuda")
    cnt = torch.tensor([4.0], dtype=torch.float32, device="cuda")
    L.check(L.lib.adam_tf_step(L.ptr(td), L.ptr(md), L.ptr(vd), L.ptr(gd), P, 1e-3, 0.9, 0.999, 1e-8, 999, L.ptr(tdev),
                               123.0, L.ptr(cnt), None, L.current_stream()), "adam")
    th, _, _ = O.adam_tf_step(np.ones(P), np.zeros(P), np.zeros(P), np.full(P, 1.5), 3, dtype=np.float64)
    np.testing.assert_allclose(td.cpu().numpy(), th, rtol=1e-6)
    # a poisoned step (non-finite loss sum: a hand-off timed out, here or on another rank) must NOT be applied
    before = (td.clone(), md.clone(), vd.clone())
    for bad in (float("nan"), float("inf")):
        loss = torch.tensor([bad], **f32)
        L.check(L.lib.adam_tf_step(L.ptr(td), L.ptr(md), L.ptr(vd), L.ptr(gd), P, 1e-3, 0.9, 0.999, 1e-8, 4, None, 1.0, None,
                                   L.ptr(loss), L.current_stream()), "adam")
    assert all(torch.equal(a, b) for a, b in zip(before, (td, md, vd)))
    loss = torch.tensor([12.5], **f32)
    L.check(L.lib.adam_tf_step(L.ptr(td), L.ptr(md), L.ptr(vd), L.ptr(gd), P, 1e-3, 0.9, 0.999, 1e-8, 4, None, 1.0, None,
                               L.ptr(loss), L.current_stream()), "adam")
    assert not torch.equal(before[0], td)


# ---------------------------------------------------------------- noise
def test_philox_noise_statistics_and_fast_mode(H):
    L = _L()
    rows, Lz, K = 1 << 14, 64, 10
    eps = torch.empty(rows, Lz, dtype=torch.float32, device="cuda")
    u = torch.empty(rows, K, dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_noise_fill(L.ptr(eps), L.ptr(u), rows, Lz, K, 0, 7, 0, None, L.current_stream()), "noise")
    e, uu = eps.cpu().numpy().astype(np.float64).ravel(), u.cpu().numpy().astype(np.float64).ravel()
    assert abs(e.mean()) < 5e-3 and abs(e.std() - 1) < 5e-3
    assert abs((e ** 3).mean()) < 2e-2 and abs((e ** 4).mean() - 3) < 5e-2
    assert uu.min() >= O.TINY_F32 and uu.max() < 1.0
    assert abs(uu.mean() - 0.5) < 3e-3 and abs(uu.var() - 1 / 12) < 2e-3
    eps2 = torch.empty(rows, Lz, dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_noise_fill(L.ptr(eps2), None, rows, Lz, K, 0, 7, 1, None, L.current_stream()), "noise")
    assert abs(np.corrcoef(e, eps2.cpu().numpy().ravel())[0, 1]) < 5e-3        # a new step is a new stream
    # fast mode of the step: eps/u = NULL -> loss within sampling distance of the parity-mode loss
    d = O.Dims(D=784, L=64, K=10, hidden=(64,))
    p = O.init_params(O.MODEL_GMVAE, d, np.random.default_rng(0))
    x, eps_h, u_h = O.make_inputs(d, 1024)
    flat = O.pack(O.MODEL_GMVAE, d, p, np.float32)
    _, t_par = H.hip_step(O.MODEL_GMVAE, d, flat, x, eps_h, u_h)
    _, t_fast = H.hip_step(O.MODEL_GMVAE, d, flat, x, None, None, seed=3, step=5)
    assert abs(t_fast[0] - t_par[0]) / abs(t_par[0]) < 5e-3
    _, t_fast2 = H.hip_step(O.MODEL_GMVAE, d, flat, x, None, None, seed=3, step=5)
    assert t_fast2[0] == t_fast[0]                                         # same (seed, step) -> same bits


@pytest.mark.parametrize("rows,Lz,K,row_base", [(64, 64, 10, 0), (33, 5, 3, 7), (16, 8, 64, (1 << 32) + 5), (100, 2, 1, 1024)])
def test_noise_fill_matches_cpu_restatement(H, rows, Lz, K, row_base):
    """gmvae_noise_fill against oracle.noise (the NumPy Philox4x32-10, pinned by the Random123 known-answer vectors in
    tests/test_pipeline.py): the uniforms bit for bit, the Box-Muller normals to the accuracy of the hardware
    log2/sin/cos; and the stream is keyed by the GLOBAL row, not by the position in the buffer."""
    L = _L()
    eps = torch.full((rows, Lz), float("nan"), dtype=torch.float32, device="cuda")
    u = torch.full((rows, K), float("nan"), dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_noise_fill(L.ptr(eps), L.ptr(u), rows, Lz, K, row_base, 0xDEADBEEF12345, 9, None, L.current_stream()), "noise")
    e_ref, u_ref = O.noise(rows, Lz, K, row_base, 0xDEADBEEF12345, 9)
    assert np.array_equal(u.cpu().numpy(), u_ref)
    np.testing.assert_allclose(eps.cpu().numpy(), e_ref, rtol=0, atol=2e-5)
    # the second half drawn on its own with the row offset = the second half of the whole
    h = rows // 2
    eps2 = torch.empty(rows - h, Lz, dtype=torch.float32, device="cuda")
    u2 = torch.empty(rows - h, K, dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_noise_fill(L.ptr(eps2), L.ptr(u2), rows - h, Lz, K, row_base + h, 0xDEADBEEF12345, 9, None,
                                   L.current_stream()), "noise")
    assert torch.equal(eps2, eps[h:]) and torch.equal(u2, u[h:])


def test_cluster_acc_kernel(H):
    L = _L()
    rng = np.random.default_rng(0)
    B, K = 1000, 10
    logits = rng.normal(size=(B, K)).astype(np.float32)
    labels = rng.integers(0, 10, B)
    ld, lab = H.dev(logits), H.dev(labels, torch.int64)
    scratch = torch.zeros(K * 10 + B, dtype=torch.int32, device="cuda")
    acc = torch.zeros(1, dtype=torch.float32, device="cuda")
    L.check(L.lib.gmvae_cluster_acc(L.ptr(ld), L.ptr(lab), B, K, 10, L.ptr(scratch), L.ptr(acc), L.current_stream()),
            "cluster_acc")
    # integer work: the [K, n_labels] histogram bit for bit; the accuracy (invariant to the mode tie-break) to fp32 rounding
    want_hist = np.zeros((K, 10), np.int32)
    np.add.at(want_hist, (logits.argmax(1), labels), 1)
    assert np.array_equal(scratch[:K * 10].cpu().numpy().reshape(K, 10), want_hist)
    assert np.array_equal(scratch[K * 10:].cpu().numpy(), logits.argmax(1))
    assert acc.item() == pytest.approx(O.cluster_acc(logits, labels, K), abs=1e-6)
    assert acc.item() == pytest.approx(O.cluster_acc_from_hist(want_hist), abs=1e-6)


SKINNY_CASES = [
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,)), 64),                                   # bin/run_train.sh:3-14
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,)), 50),                                   # ragged last row tile
    ("gmvae", O.Dims(D=256, L=32, K=7, hidden=(128,), temperature=0.6), 100),                   # other widths, 7 row tiles
    ("gmvae", O.Dims(D=400, L=16, K=16, hidden=(64,), sigma_min=0.8, raw_sigma_bias=0.25, gen_bias_init=-0.4), 9),  # clamp active
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(192,)), 128),
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(256,)), 600),                                  # 38 row tiles, ragged; 10 row chunks in W
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,)), 1024),                                 # the largest batch it takes by default
    # latent sizes that are no multiple of 16 (scripts/run_gmvae.py:17 default latent_size = 8): ragged latent tiles, masked k-groups
    ("gmvae", O.Dims(D=784, L=8, K=10, hidden=(256,)), 64),
    ("gmvae", O.Dims(D=784, L=20, K=10, hidden=(128,)), 40),
    ("gmvae", O.Dims(D=256, L=4, K=5, hidden=(64,)), 33),
    ("vae", O.Dims(D=784, L=8, K=1, hidden=(512,)), 100),
    ("vae", O.Dims(D=784, L=24, K=1, hidden=(128,)), 70),
    # the VAE with the standard-normal prior (scripts/vae.py:167-185): eight launches, no y path
    ("vae", O.Dims(D=784, L=128, K=1, hidden=(512,)), 64),
    ("vae", O.Dims(D=784, L=16, K=1, hidden=(512,)), 100),                                       # BASELINE configs[0] at H = 512, ragged
    ("vae", O.Dims(D=256, L=32, K=1, hidden=(128,), sigma_min=0.7, raw_sigma_bias=0.25, gen_bias_init=0.3), 37),
    ("vae", O.Dims(D=784, L=64, K=1, hidden=(256,)), 512),
    # VAE_GMP: the learned mixture prior's log-density, its share of dz and its variables' gradients as row kernels between the
    # skinny launches (eleven launches), the variables' update in the loss-tail workgroup
    ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(512,)), 256),                                  # BASELINE configs[1] at H = 512
    ("vae_gmp", O.Dims(D=784, L=128, K=10, hidden=(512,)), 64),
    ("vae_gmp", O.Dims(D=256, L=20, K=7, hidden=(128,), sigma_min=0.6, raw_sigma_bias=0.25), 45),
    ("vae_gmp", O.Dims(D=400, L=16, K=80, hidden=(64,)), 30),                                    # K > 64: the tiled mixture log-density
    # above 128 rows: groups of 2 / 4 row tiles per workgroup, 64-column tiles for the D-wide layers, [64 x 64] W tiles (one launch,
    # or two batch shares + the optimizer launch), two rows per workgroup on the y path -- ragged in rows, columns and shares
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(512,)), 1000),                                   # BASELINE configs[2] at H = 512, ragged
    ("gmvae", O.Dims(D=400, L=20, K=7, hidden=(512,), temperature=0.7), 900),                    # ragged 64-column tile, W in one launch
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(1024,)), 520),                                  # H = 1024: W in two launches, ragged shares
    ("vae_gmp", O.Dims(D=784, L=64, K=10, hidden=(512,)), 530),
    ("vae", O.Dims(D=784, L=16, K=1, hidden=(320,)), 700),
]


def _random_skinny_cases(n, seed):
    """Seeded random shapes INSIDE the skinny schedule's gates (one hidden layer of 64 .. 320 in steps of 64, latent sizes that
    are multiples of 4, K <= 16 for the GMVAE, batches from one row to a few ragged row tiles), all three models."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        model = ["gmvae", "vae", "vae_gmp"][i % 3]
        d = O.Dims(D=int(rng.choice([64, 112, 256, 400, 784])), L=int(rng.choice([4, 8, 12, 16, 28, 36, 64])),
                   K=1 if model == "vae" else int(rng.choice([2, 3, 7, 10, 16])), hidden=(int(rng.choice([64, 128, 192, 320])),),
                   temperature=float(rng.choice([1.0, 0.7])), sigma_min=float(rng.choice([0.0, 0.0, 0.5])))
        out.append((model, d, int(rng.choice([1, 5, 16, 31, 77, 150]))))
    return out


RANDOM_SKINNY_CASES = _random_skinny_cases(15, 20261005)


def _random_large_skinny_cases(n, seed):
    """Seeded random shapes for the schedule's forms ABOVE 128 rows: row-tile groups (ragged last group), 64-column tiles with a
    ragged last tile (D no multiple of 64), the y path's 1 / 2 / 4 rows per workgroup, all three weight-gradient forms (one
    workgroup per tile with bf16 pieces or fp32 instructions; batch shares with a last-arriver optimizer), the mixture prior's
    one-launch form with ragged strips."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        model = ["gmvae", "vae_gmp", "vae"][i % 3]
        d = O.Dims(D=int(rng.choice([208, 400, 784, 1008])), L=int(rng.choice([8, 20, 32, 64, 128])),
                   K=1 if model == "vae" else int(rng.choice([3, 7, 10, 16])), hidden=(int(rng.choice([128, 256, 320, 512, 768])),),
                   temperature=float(rng.choice([1.0, 0.7])), sigma_min=float(rng.choice([0.0, 0.0, 0.5])))
        out.append((model, d, int(rng.choice([130, 257, 333, 520, 777, 1025, 1290, 1700, 2100]))))
    return out


# (the seed is the first after 20261006 whose 12 shapes take EVERY form: three batch-share launches -- one of them VAE_GMP --, one
#  fp32 one-workgroup-per-tile launch, 1 / 2 / 4 rows per y-path workgroup; a coverage choice, made before any shape was run)
RANDOM_LARGE_SKINNY_CASES = _random_large_skinny_cases(12, 20261026)


@pytest.mark.parametrize("name,d,B", RANDOM_LARGE_SKINNY_CASES,
                         ids=[f"{n}-D{d.D}-L{d.L}-K{d.K}-H{d.hidden[0]}-B{B}" for n, d, B in RANDOM_LARGE_SKINNY_CASES])
def test_skinny_schedule_above_128_rows_on_random_shapes(H, monkeypatch, name, d, B):
    monkeypatch.setenv("GMVAE_NO_MEGA", "1")
    monkeypatch.setenv("GMVAE_NO_FUSED", "1")
    model = O.MODEL_NAMES[name]
    assert _L().step_schedule(H.dims_of(d, B), model) == "skinny"
    rng = np.random.default_rng(B * 13 + d.D + d.L)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)


def test_skinny_schedule_four_rows_per_y_path_workgroup(H, monkeypatch):
    """Above 1536 rows the y path carries four rows per workgroup; here the last workgroup's rows are ragged.  (The schedule's
    batch bound is 4096 rows: beyond, the general schedule.)"""
    monkeypatch.setenv("GMVAE_NO_MEGA", "1")
    name, d, B = "gmvae", O.Dims(D=256, L=32, K=10, hidden=(256,)), 1598
    assert _L().step_schedule(H.dims_of(d, 4097), O.MODEL_NAMES[name]) == "general"
    model = O.MODEL_NAMES[name]
    assert _L().step_schedule(H.dims_of(d, B), model) == "skinny"
    rng = np.random.default_rng(B)
    p = O.init_params(model, d, rng)
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)


@pytest.mark.parametrize("name,d,B", RANDOM_SKINNY_CASES, ids=[f"{n}-D{d.D}-L{d.L}-K{d.K}-H{d.hidden[0]}-B{B}" for n, d, B in RANDOM_SKINNY_CASES])
def test_skinny_schedule_on_random_shapes(H, monkeypatch, name, d, B):
    """The skinny schedule (forced where the mega schedule would take H = 64) against the oracle on random eligible shapes."""
    monkeypatch.setenv("GMVAE_NO_MEGA", "1")
    monkeypatch.setenv("GMVAE_NO_FUSED", "1")
    L = _L()
    model = O.MODEL_NAMES[name]
    assert L.step_schedule(H.dims_of(d, B), model) == "skinny"
    rng = np.random.default_rng(B * 17 + d.D + d.L)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)


@pytest.mark.parametrize("name,d,B", SKINNY_CASES, ids=[f"{n}-D{d.D}-L{d.L}-K{d.K}-H{d.hidden[0]}-B{B}" for n, d, B in SKINNY_CASES])
def test_skinny_schedule_matches_oracle(H, monkeypatch, name, d, B):
    """The small-batch / wide-layer schedule (csrc/skinny.hpp: 10 launches, register-direct tiles) against the oracle --
    forced where the mega schedule would otherwise take the sizes (H = 64) -- and NOT bit-identical to the general
    schedule (evidence that it ran); with external noise and with the in-kernel Philox draw."""
    monkeypatch.setenv("GMVAE_NO_MEGA", "1")
    monkeypatch.setenv("GMVAE_NO_FUSED", "1")
    model = O.MODEL_NAMES[name]
    rng = np.random.default_rng(B)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u)
    flat = O.pack(model, d, p, np.float32)
    g_sk, t_sk = H.hip_step(model, d, flat, x, eps, u)
    monkeypatch.setenv("GMVAE_NO_SKINNY", "1")
    g_gen, t_gen = H.hip_step(model, d, flat, x, eps, u)
    lay, _, _ = O.param_layout(model, d)
    same = True
    for nm, shape, off in lay:                      # (the alignment padding between tensors is never written by this schedule)
        n = int(np.prod(shape))
        same = same and np.array_equal(g_sk[off:off + n], g_gen[off:off + n])
        np.testing.assert_allclose(g_sk[off:off + n], g_gen[off:off + n], rtol=0, atol=2e-4 * np.abs(g_gen[off:off + n]).max(), err_msg=nm)
    assert not same
    np.testing.assert_allclose(t_sk[:5], t_gen[:5], rtol=2e-6)
