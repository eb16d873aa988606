"""NumPy restatement of the reference hot path (test infrastructure, see __init__).

Every function cites the reference lines it follows (paths are relative to the
upstream repo's ``scripts/`` directory).  ``dtype`` selects fp64 (truth) or
fp32 (stand-in for the reference's CPU TF path).  Noise (``eps``, ``u``) and
parameters are explicit inputs: TF's Philox streams cannot be reproduced.

Row convention for the IWAE extension (not in the
reference): sample-dependent tensors have R = B*S rows, row r = b*S + s.
At S == 1 everything below is the reference's single-sample ELBO.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MODEL_VAE = 0
MODEL_VAE_GMP = 1
MODEL_GMVAE = 2
MODEL_NAMES = {"vae": MODEL_VAE, "vae_gmp": MODEL_VAE_GMP, "gmvae": MODEL_GMVAE}

LOG_2PI = math.log(2.0 * math.pi)
TINY_F32 = float(np.finfo(np.float32).tiny)  # RelaxedOneHotCategorical uniform minval

__all__ = [
    "MODEL_VAE", "MODEL_VAE_GMP", "MODEL_GMVAE", "MODEL_NAMES", "Dims",
    "param_specs", "param_layout", "init_params", "pack", "unpack",
    "softplus", "sigmoid", "forward", "loss_and_grads", "adam_tf_step",
    "train_step", "make_inputs", "flops_per_step", "cluster_acc", "TINY_F32",
    "philox4x32_10", "binarize", "cluster_acc_from_hist", "noise",
]


@dataclass
class Dims:
    """Problem sizes + the hyper-parameters runners.py:78-101 binds."""
    D: int = 784
    L: int = 8                       # run_gmvae.py:17 latent_size
    K: int = 10                      # run_gmvae.py:23 mixture_components
    hidden: Sequence[int] = (64,)    # [hidden_size]*num_layers, runners.py:83
    S: int = 1                       # IWAE samples (build extension)
    sigma_min: float = 0.0           # runners.py:84
    raw_sigma_bias: float = 0.5      # runners.py:85
    temperature: float = 1.0         # runners.py:86
    gen_bias_init: object = 0.0      # gmvae.py:285 / vae.py:199; scalar, or a vector [D] (base.py:102-103 "scalar or vector Tensor")
    act: str = "relu"                # hidden_activation_fn of every conditional's MLP (base.py:19,90,153; gmvae.py:282, vae.py:196):
                                     # "relu" (the reference's default: tf.nn.relu), "tanh", "sigmoid", "elu"


# --------------------------------------------------------------------------
# parameters (SURVEY.md A.1 creation order; names = TF variable names, 5.4)
# --------------------------------------------------------------------------
def _mlp_specs(name: str, n_in: int, hidden: Sequence[int], n_out: int):
    dims = [n_in] + list(hidden) + [n_out]
    out = []
    for i in range(len(dims) - 1):
        out.append((f"{name}_fcnet/linear_{i}/w", (dims[i], dims[i + 1])))
        out.append((f"{name}_fcnet/linear_{i}/b", (dims[i + 1],)))
    return out


def param_specs(model: int, d: Dims) -> List[Tuple[str, Tuple[int, ...]]]:
    h = list(d.hidden)
    if model == MODEL_GMVAE:
        # gmvae.py:238,243,246,251 -- variables are created at first call.
        return (_mlp_specs("encoder_y", d.D, h, d.K)
                + _mlp_specs("prior_gmm", d.K, [], 2 * d.L)      # gmvae.py:321-327
                + _mlp_specs("encoder_gmm", d.D + d.K, h, 2 * d.L)
                + _mlp_specs("decoder", d.L, h, d.D))
    specs = []
    if model == MODEL_VAE_GMP:
        specs += [("loc", (d.K, d.L)), ("raw_scale_diag", (d.K, d.L)),
                  ("mixture_logits", (d.K,))]                   # vae.py:233-238
    return specs + _mlp_specs("encoder", d.D, h, 2 * d.L) + _mlp_specs("decoder", d.L, h, d.D)


def param_layout(model: int, d: Dims, align: int = 4):
    """Flat fp32 buffer layout: every tensor starts on a 16-byte boundary."""
    off, out, real = 0, [], 0
    for name, shape in param_specs(model, d):
        n = int(np.prod(shape))
        out.append((name, shape, off))
        real += n
        off += (n + align - 1) // align * align
    return out, off, real


def init_params(model: int, d: Dims, rng: np.random.Generator, dtype=np.float64) -> Dict[str, np.ndarray]:
    """base.py:12 Xavier-uniform weights / zero biases; tf.get_variable default
    (Glorot-uniform) for the VAE_GMP prior variables (vae.py:233-238)."""
    p = {}
    for name, shape in param_specs(model, d):
        if name.endswith("/b"):
            p[name] = np.zeros(shape, dtype)
        else:
            fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[0], shape[0])
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            p[name] = rng.uniform(-lim, lim, size=shape).astype(dtype)
    return p


def pack(model: int, d: Dims, params: Dict[str, np.ndarray], dtype=np.float32) -> np.ndarray:
    lay, P, _ = param_layout(model, d)
    flat = np.zeros(P, dtype)
    for name, shape, off in lay:
        flat[off:off + int(np.prod(shape))] = np.asarray(params[name], dtype).ravel()
    return flat


def unpack(model: int, d: Dims, flat: np.ndarray) -> Dict[str, np.ndarray]:
    lay, _, _ = param_layout(model, d)
    return {name: np.array(flat[off:off + int(np.prod(shape))]).reshape(shape) for name, shape, off in lay}


# --------------------------------------------------------------------------
# elementwise pieces
# --------------------------------------------------------------------------
def softplus(x):
    """Overflow-safe tf.nn.softplus: max(x,0) + log1p(exp(-|x|))."""
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def sigmoid(x):
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1.0 / (1.0 + e), e / (1.0 + e))


def _log_softmax(a):
    m = a.max(axis=-1, keepdims=True)
    s = a - m
    return s - np.log(np.exp(s).sum(axis=-1, keepdims=True))


ACTS = ("relu", "tanh", "sigmoid", "elu")


def _act(h, act):
    """tf.nn.relu / tf.tanh / tf.sigmoid / tf.nn.elu (alpha = 1) on a pre-activation."""
    if act == "relu":
        return np.maximum(h, 0)
    if act == "tanh":
        return np.tanh(h)
    if act == "sigmoid":
        return sigmoid(h)
    if act == "elu":
        return np.where(h > 0, h, np.expm1(np.minimum(h, 0)))
    raise ValueError(act)


def _dact(a, act):
    """f'(pre) as a function of the ACTIVATION a = f(pre) (what the backward pass keeps)."""
    if act == "relu":
        return (a > 0).astype(a.dtype)
    if act == "tanh":
        return 1 - a * a
    if act == "sigmoid":
        return a * (1 - a)
    if act == "elu":
        return np.where(a > 0, 1.0, a + 1.0).astype(a.dtype)
    raise ValueError(act)


def _mlp_fwd(p, name, n_layers, x, pres=None, act="relu"):
    """snt.nets.MLP, activate_final=False, relu hidden (base.py:47-60).
    Returns output and the list of layer inputs [h_0 .. h_n]; `pres` (a list,
    optional) receives (pre-activation, sum_k |input_k| |w_kj| + |b_j|) of every
    hidden layer -- what the trajectory tests need to tell a pre-activation
    that is zero to within fp32 rounding from one that is not."""
    hs = [x]
    h = x
    for i in range(n_layers):
        w, b = p[f"{name}_fcnet/linear_{i}/w"], p[f"{name}_fcnet/linear_{i}/b"]
        a = h
        h = a @ w + b
        if i < n_layers - 1:
            if pres is not None:
                pres.append((h, np.abs(a) @ np.abs(w) + np.abs(b)))
            h = _act(h, act)
            hs.append(h)
    return h, hs


def _mlp_bwd(p, g, name, n_layers, hs, dout, need_dx=True, masks=None, act="relu"):
    """Closed-form backward of _mlp_fwd.  hs[i] is the input
    of layer i; hs[i>0] is post-ReLU so the mask is hs[i] > 0 -- unless
    `masks[i]` (bool, same shape) names the subgradient to take: at a
    pre-activation of exactly zero TF's ReluGrad takes 0, but an fp32 and an
    fp64 evaluation of a pre-activation that is zero to within rounding can
    land on different sides; the trajectory tests pass the device's choice."""
    d = dout
    for i in reversed(range(n_layers)):
        g[f"{name}_fcnet/linear_{i}/w"] = hs[i].T @ d
        g[f"{name}_fcnet/linear_{i}/b"] = d.sum(axis=0)
        if i > 0 or need_dx:
            d = d @ p[f"{name}_fcnet/linear_{i}/w"].T
            if i > 0:
                if masks is not None and i < len(masks) and masks[i] is not None:
                    assert act == "relu", "subgradient masks are a ReLU matter"
                    d = d * masks[i]
                else:
                    d = d * _dact(hs[i], act)
    return d


def _normal_head(out, L, c, smin):
    """base.py:66-72: split, sigma = max(softplus(raw + c), sigma_min)."""
    mu, raw = out[:, :L], out[:, L:]
    sp = softplus(raw + c)
    return mu, np.maximum(sp, smin), raw


def _mvn_logprob(z, mu, sigma):
    """MultivariateNormalDiag.log_prob (A7): computed from z, not eps."""
    e = (z - mu) / sigma
    return (-0.5 * e * e - 0.5 * LOG_2PI).sum(axis=1) - np.log(sigma).sum(axis=1)


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
def forward(model: int, d: Dims, p: Dict[str, np.ndarray], x: np.ndarray,
            eps: np.ndarray, u: Optional[np.ndarray] = None, dtype=np.float64):
    """gmvae.py:238-267 / vae.py:167-185.  x: bool/uint8 [B,D]; eps [B*S,L];
    u [B*S,K] (GMVAE only).  Returns a cache dict (all intermediates)."""
    p = {k: np.asarray(v, dtype) for k, v in p.items()}
    xf = np.asarray(x).astype(dtype)                       # gmvae.py:86,104 / vae.py:75
    B, S, L, K = x.shape[0], d.S, d.L, d.K
    R = B * S
    nl = len(d.hidden) + 1
    c, smin = dtype(d.raw_sigma_bias), dtype(d.sigma_min)
    eps = np.asarray(eps, dtype).reshape(R, L)
    xr = np.repeat(xf, S, axis=0) if S > 1 else xf         # row r = b*S + s
    C: Dict[str, object] = {"B": B, "R": R, "xf": xf, "xr": xr, "eps": eps}

    if model == MODEL_GMVAE:
        T = dtype(d.temperature)
        u = np.asarray(u, dtype).reshape(R, K)
        pre_y, pre_g = [], []
        logits, hs_y = _mlp_fwd(p, "encoder_y", nl, xf, pre_y, act=d.act)     # gmvae.py:238
        g = -np.log(-np.log(u))                                    # A9 Gumbel
        a = (np.repeat(logits, S, axis=0) + g) / T
        y = np.exp(_log_softmax(a))                                # gmvae.py:240
        lnpi = _log_softmax(logits)
        pi = np.exp(lnpi)
        nent_b = (pi * lnpi).sum(axis=1)                           # gmvae.py:262, utils.py:165-170
        pp = y @ p["prior_gmm_fcnet/linear_0/w"] + p["prior_gmm_fcnet/linear_0/b"]  # gmvae.py:243
        mu_p, sig_p, raw_p = _normal_head(pp, L, c, smin)
        qp, hs_g = _mlp_fwd(p, "encoder_gmm", nl, np.concatenate([xr, y], axis=1), pre_g, act=d.act)  # gmvae.py:246, base.py:66
        C["pre"] = {"encoder_y": pre_y, "encoder_gmm": pre_g}
        C.update(logits=logits, hs_y=hs_y, y=y, pi=pi, lnpi=lnpi, nent_b=nent_b, pp=pp,
                 mu_p=mu_p, sig_p=sig_p, raw_p=raw_p, hs_g=hs_g, gumbel=g)
        enc_name = "encoder_gmm"
    else:
        pre_e = []
        qp, hs_e = _mlp_fwd(p, "encoder", nl, xf, pre_e, act=d.act)           # vae.py:170
        C["pre"] = {"encoder": pre_e}
        if S > 1:
            qp = np.repeat(qp, S, axis=0)
        C.update(hs_e=hs_e)
        nent_b = np.zeros(B, dtype)
        enc_name = "encoder"
    mu_q, sig_q, raw_q = _normal_head(qp, L, c, smin)
    z = mu_q + sig_q * eps                                         # A6, gmvae.py:248 / vae.py:171
    logq = _mvn_logprob(z, mu_q, sig_q)

    if model == MODEL_GMVAE:
        logp = _mvn_logprob(z, mu_p, sig_p)                        # gmvae.py:258
    elif model == MODEL_VAE:
        logp = (-0.5 * z * z - 0.5 * LOG_2PI).sum(axis=1)          # vae.py:247-250
    else:                                                          # vae.py:240-244 MixtureSameFamily
        loc, s = p["loc"], softplus(p["raw_scale_diag"])
        lnw = _log_softmax(p["mixture_logits"])
        t = (z[:, None, :] - loc[None]) / s[None]                  # [R,K,L]
        lnN = (-0.5 * t * t - 0.5 * LOG_2PI).sum(axis=2) - np.log(s).sum(axis=1)[None]
        comp = lnw[None] + lnN
        m = comp.max(axis=1, keepdims=True)
        logp = (m + np.log(np.exp(comp - m).sum(axis=1, keepdims=True)))[:, 0]
        C.update(resp=np.exp(comp - logp[:, None]), gmp_t=t, gmp_s=s, lnw=lnw)

    pre_d = []
    lam, hs_d = _mlp_fwd(p, "decoder", nl, z, pre_d, act=d.act)               # gmvae.py:251 / vae.py:174
    C["pre"]["decoder"] = pre_d
    lam = lam + np.asarray(d.gen_bias_init, dtype)                 # base.py:135 (scalar or [D] vector, broadcast over rows)
    logpx = (xr * lam - softplus(lam)).sum(axis=1)                 # A8, gmvae.py:254

    nent_r = np.repeat(nent_b, S) if S > 1 else nent_b
    logw = logpx + logp - logq - nent_r                            # A15; S=1: -loss_b
    if S == 1:
        rw = np.ones(R, dtype)
        bound = logw
    else:
        lw = logw.reshape(B, S)
        m = lw.max(axis=1, keepdims=True)
        bound = (m[:, 0] + np.log(np.exp(lw - m).sum(axis=1))) - math.log(S)
        rw = np.exp(lw - (bound + math.log(S))[:, None]).reshape(R)
    C.update(enc_name=enc_name, qp=qp, mu_q=mu_q, sig_q=sig_q, raw_q=raw_q, z=z, logq=logq,
             logp=logp, lam=lam, hs_d=hs_d, logpx=logpx, logw=logw, rw=rw, bound=bound,
             nll=-logpx.mean(), kl=(logq - logp).mean(), nent=nent_b.mean(),
             loss=-bound.mean())
    return C


# --------------------------------------------------------------------------
# backward (SURVEY.md 8(a) A12 closed form; checked against fp64 autograd in
# tests/test_oracle.py)
# --------------------------------------------------------------------------
def loss_and_grads(model: int, d: Dims, p, x, eps, u=None, dtype=np.float64, relu_masks=None):
    """relu_masks: optional {net name: [None, mask of hidden layer 1, ...]} -- the ReLU subgradients to take instead of
    (activation > 0); see _mlp_bwd.  The forward pass is unaffected."""
    rm = relu_masks or {}
    p = {k: np.asarray(v, dtype) for k, v in p.items()}
    C = forward(model, d, p, x, eps, u, dtype)
    B, R, S, L, K = C["B"], C["R"], d.S, d.L, d.K
    nl = len(d.hidden) + 1
    c, smin = dtype(d.raw_sigma_bias), dtype(d.sigma_min)
    g: Dict[str, np.ndarray] = {}
    w = (C["rw"] / B)[:, None]                                     # per-row weight incl. batch mean
    z, mu_q, sig_q, eps_ = C["z"], C["mu_q"], C["sig_q"], C["eps"]

    dlam = w * (sigmoid(C["lam"]) - C["xr"])
    dz_dec = _mlp_bwd(p, g, "decoder", nl, C["hs_d"], dlam, masks=rm.get("decoder"), act=d.act)

    if model == MODEL_GMVAE:
        t = (z - C["mu_p"]) / C["sig_p"]
        dprior_z = w * t / C["sig_p"]
    elif model == MODEL_VAE:
        dprior_z = w * z
    else:
        r, t, s = C["resp"], C["gmp_t"], C["gmp_s"]                # [R,K], [R,K,L], [K,L]
        dprior_z = w * (r[:, :, None] * t / s[None]).sum(axis=1)
        wr = (w * r)[:, :, None]
        g["loc"] = -(wr * t / s[None]).sum(axis=0)
        ds = (wr * (1.0 - t * t) / s[None]).sum(axis=0)
        g["raw_scale_diag"] = ds * sigmoid(p["raw_scale_diag"])
        g["mixture_logits"] = -(w * (r - np.exp(C["lnw"])[None])).sum(axis=0)

    dmu_q = dz_dec + dprior_z
    dsig_q = dmu_q * eps_ - w / sig_q
    draw_q = dsig_q * sigmoid(C["raw_q"] + c) * (softplus(C["raw_q"] + c) > smin)
    dqp = np.concatenate([dmu_q, draw_q], axis=1)

    if model == MODEL_GMVAE:
        dmu_p = -w * t / C["sig_p"]
        dsig_p = w * (1.0 - t * t) / C["sig_p"]
        draw_p = dsig_p * sigmoid(C["raw_p"] + c) * (softplus(C["raw_p"] + c) > smin)
        dpp = np.concatenate([dmu_p, draw_p], axis=1)
        g["prior_gmm_fcnet/linear_0/w"] = C["y"].T @ dpp
        g["prior_gmm_fcnet/linear_0/b"] = dpp.sum(axis=0)
        dxy = _mlp_bwd(p, g, "encoder_gmm", nl, C["hs_g"], dqp, masks=rm.get("encoder_gmm"), act=d.act)    # [R, D+K]
        dy = dxy[:, d.D:] + dpp @ p["prior_gmm_fcnet/linear_0/w"].T
        y = C["y"]
        da = y * (dy - (y * dy).sum(axis=1, keepdims=True))
        dl = (da / dtype(d.temperature)).reshape(B, S, K).sum(axis=1)
        pi, lnpi = C["pi"], C["lnpi"]
        dl = dl + (pi * (lnpi - C["nent_b"][:, None])) / B
        _mlp_bwd(p, g, "encoder_y", nl, C["hs_y"], dl, need_dx=False, masks=rm.get("encoder_y"), act=d.act)
    else:
        dq_b = dqp.reshape(B, S, 2 * L).sum(axis=1) if S > 1 else dqp
        _mlp_bwd(p, g, "encoder", nl, C["hs_e"], dq_b, need_dx=False, masks=rm.get("encoder"), act=d.act)
    C["dlam"], C["dqp"] = dlam, dqp
    return C, g


# --------------------------------------------------------------------------
# optimiser: TF1 AdamOptimizer (runners.py:181-183), SURVEY.md A13
# --------------------------------------------------------------------------
def adam_tf_step(theta, m, v, grad, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, dtype=np.float32):
    """t is the 1-based step count AFTER increment.  eps is added to the
    UN-corrected sqrt(v) (torch.optim.Adam differs).  Written in the form of
    TF's ApplyAdam functor (tensorflow/core/kernels/training_ops.cc):
    m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= m*alpha/(sqrt(v)+eps), with the
    hyper-parameters held in `dtype` like TF holds them in T (so in fp32
    1-b2 = 1 - 0.999f, not fp32(0.001))."""
    dt = dtype
    fb1, fb2 = dt(b1), dt(b2)
    alpha = dt(float(lr) * math.sqrt(1.0 - float(fb2) ** t) / (1.0 - float(fb1) ** t))
    m = m + (grad - m) * (dt(1) - fb1)
    v = v + (grad * grad - v) * (dt(1) - fb2)
    theta = theta - m * alpha / (np.sqrt(v) + dt(eps))
    return theta.astype(dt), m.astype(dt), v.astype(dt)


def train_step(model, d, flat, m, v, t, x, eps, u=None, lr=1e-3, dtype=np.float32, relu_masks=None):
    """One full reference step on the flat buffer: fwd + bwd + TF-Adam."""
    p = unpack(model, d, flat)
    C, g = loss_and_grads(model, d, p, x, eps, u, dtype, relu_masks=relu_masks)
    gflat = pack(model, d, g, dtype)
    flat, m, v = adam_tf_step(flat.astype(dtype), m, v, gflat, t, lr=lr, dtype=dtype)
    return flat, m, v, C, gflat


# --------------------------------------------------------------------------
# synthetic inputs (SURVEY.md 8(d) / BASELINE.md section 4)
# --------------------------------------------------------------------------
def make_inputs(d: Dims, B: int, model: int = MODEL_GMVAE, seed_x=1234, seed_noise=42):
    x = (np.random.default_rng(seed_x).random((B, d.D)) < 0.87).astype(np.uint8)
    rn = np.random.default_rng(seed_noise)
    eps = rn.standard_normal((B * d.S, d.L)).astype(np.float32)
    u = rn.uniform(TINY_F32, 1.0, (B * d.S, d.K)).astype(np.float32)
    u = np.clip(u, TINY_F32, np.nextafter(np.float32(1), np.float32(0)))
    return x, eps, (u if model == MODEL_GMVAE else None)


def flops_per_step(model: int, d: Dims, B: int) -> float:
    """SURVEY.md A.1 FLOP rule: 2*(fwd + dW + dX MACs), x-input dX excluded."""
    def mac(n_in, n_out):
        dims = [n_in] + list(d.hidden) + [n_out]
        return sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    h0, S = d.hidden[0], d.S
    if model == MODEL_GMVAE:
        fwd = mac(d.D, d.K) + S * (d.K * 2 * d.L + mac(d.D + d.K, 2 * d.L) + mac(d.L, d.D))
        dx = fwd - d.D * h0 * (1 + S)
    else:
        fwd = mac(d.D, 2 * d.L) + S * mac(d.L, d.D)
        dx = fwd - d.D * h0
    return 2.0 * B * (2 * fwd + dx)


def philox4x32_10(c: np.ndarray, k0: int, k1: int) -> np.ndarray:
    """Philox4x32-10 (Salmon et al., SC'11) on an array of counters c[..., 4] (uint32) with key (k0, k1):
    the generator of the HIP path's noise streams (gmvae_amd/csrc/aux.hpp), restated with NumPy integers."""
    c = [c[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK
        n3 = p0 & MASK
        c = [n0, n1, n2, n3]
        k0 = (k0 + np.uint64(0x9E3779B9)) & MASK
        k1 = (k1 + np.uint64(0xBB67AE85)) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def binarize(pixels: np.ndarray, rows: np.ndarray, seed: int, step: int, out_row0: int = 0) -> np.ndarray:
    """scripts/runners.py:44-47 `_preprocess` for one batch: image = float32(pixel) / 255; x = image < uniform.
    The uniforms are the HIP path's: Philox4x32-10, counter = (quad index in the GLOBAL batch, 0x40000000, step),
    key = seed, u = (bits >> 8) * 2^-24; out_row0 = global index of this batch's first row (a data-parallel shard).
    pixels uint8 [N, D], rows int [B] -> uint8 [B, D] of 0/1 (bit-exact contract)."""
    B, D = len(rows), pixels.shape[1]
    q = np.arange(B * D // 4, dtype=np.uint64) + np.uint64(out_row0 * (D // 4))
    c = np.stack([q & np.uint64(0xFFFFFFFF), (q >> np.uint64(32)) | np.uint64(0x40000000),
                  np.full_like(q, step & 0xFFFFFFFF), np.full_like(q, (step >> 32) & 0xFFFFFFFF)], axis=-1).astype(np.uint32)
    r = philox4x32_10(c, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = ((r >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(B, D)
    img = pixels[np.asarray(rows)].astype(np.float32) / np.float32(255.0)
    return (img < u).astype(np.uint8)


def noise(rows: int, L: int, K: int, row_base: int, seed: int, step: int) -> Tuple[np.ndarray, np.ndarray]:
    """The HIP path's in-kernel noise (gmvae_amd/csrc/aux.hpp noise_vals, the stand-in for tf.random_normal /
    tf.random_uniform inside the TFP samplers at gmvae.py:240,248 and vae.py:171) restated on the CPU:
    Philox4x32-10, counter = (global row, quad of the row | stream bit, step), key = seed.
    u [rows, K] = max((bits >> 8) * 2^-24, tiny) is bit-exact; eps [rows, L] is Box-Muller,
    sqrt(-2 ln(1 - u0)) * (cos, sin)(2 pi u1), which the GPU evaluates with its hardware log2/sin/cos (agreement to a
    few 1e-6 absolute, not bitwise)."""
    def bits(n_cols, stream):
        qpr = (n_cols + 3) // 4
        row = (np.arange(rows, dtype=np.uint64) + np.uint64(row_base))[:, None].repeat(qpr, 1)
        quad = np.arange(qpr, dtype=np.uint64)[None, :].repeat(rows, 0)
        c1 = (quad & np.uint64(0x00FFFFFF)) | (((row >> np.uint64(32)) & np.uint64(0x3F)) << np.uint64(24)) | np.uint64(stream)
        c = np.stack([row & np.uint64(0xFFFFFFFF), c1, np.full_like(row, step & 0xFFFFFFFF),
                      np.full_like(row, (step >> 32) & 0xFFFFFFFF)], axis=-1).astype(np.uint32)
        r = philox4x32_10(c, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)          # [rows, qpr, 4]
        return (r >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
    ub = bits(K, 0x80000000).reshape(rows, -1)[:, :K]
    u = np.maximum(ub.astype(np.float32), np.float32(TINY_F32))
    eb = bits(L, 0)
    rad = np.sqrt(-2.0 * np.log(1.0 - eb[..., 0::2]))
    ang = 2.0 * np.pi * eb[..., 1::2]
    e = np.stack([rad * np.cos(ang), rad * np.sin(ang)], axis=-1)                     # [rows, qpr, 2 pairs, (cos, sin)]
    eps = e.reshape(rows, -1)[:, :L].astype(np.float32)
    return eps, u


def cluster_acc_from_hist(hist: np.ndarray) -> float:
    """utils.cluster_acc (scripts/utils.py:173-191) from the [K, n_labels] cluster x label histogram alone: every
    sample of a cluster whose label is the cluster's majority label is a match, so matches = sum_k max_l hist[k, l].
    This is the form that data-parallel ranks can combine (sum the histograms, then reduce)."""
    hist = np.asarray(hist, dtype=np.int64)
    n = hist.sum()
    return float(hist.max(axis=1).sum()) / float(n) if n else 0.0


def cluster_acc(logits: np.ndarray, labels: np.ndarray, K: int) -> float:
    """utils.py:173-191 (with utils.py:156-162 mode_tensor): argmax cluster ->
    majority label -> match rate.  Ties in the mode: tf.unique_with_counts
    keeps first-occurrence order and argmax takes the first maximum."""
    preds = np.argmax(logits, axis=1)
    real = np.zeros(len(preds), np.float32)
    for k in range(K):
        idx = preds == k
        lab = labels[idx]
        if lab.size == 0:
            mode = 0.0
        else:
            uniq, first, cnt = np.unique(lab, return_index=True, return_counts=True)
            order = np.argsort(first, kind="stable")
            mode = float(uniq[order][np.argmax(cnt[order])])
        real += idx.astype(np.float32) * mode
    return float(np.mean(real == labels.astype(np.float32)))
