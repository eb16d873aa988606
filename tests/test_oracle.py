"""Pins the CPU oracle: closed-form KATs, an independent
torch.distributions + autograd fp64 statement, TF-Adam formula, fp32-vs-fp64."""
import math

import numpy as np
import pytest
import torch

import oracle as O

F64 = torch.float64      # never change the global default dtype: other test modules share the process


def zero_params(model, d):
    return {n: np.zeros(s) for n, s in O.param_specs(model, d)}


# ------------------------------------------------------------------ KATs
@pytest.mark.parametrize("D,K,expect", [(784, 10, 541.124804), (3072, 64, 2125.189256)])
def test_kat_gmvae_zero_weights(D, K, expect):
    d = O.Dims(D=D, L=64, K=K, hidden=(64,))
    x, eps, u = O.make_inputs(d, 8)
    C = O.forward(O.MODEL_GMVAE, d, zero_params(O.MODEL_GMVAE, d), x, eps, u)
    assert C["loss"] == pytest.approx(D * math.log(2) - math.log(K), abs=1e-9)
    assert C["loss"] == pytest.approx(expect, abs=1e-5)
    assert C["kl"] == pytest.approx(0.0, abs=1e-12)
    assert C["nent"] == pytest.approx(-math.log(K), abs=1e-12)


@pytest.mark.parametrize("L,e,kl", [(2, 0.0, 0.05252988), (2, 1.0, 0.00135585), (64, 1.0, 0.04338719)])
def test_kat_vae_zero_weights(L, e, kl):
    d = O.Dims(D=784, L=L, K=1, hidden=(64,))
    x, _, _ = O.make_inputs(d, 4, O.MODEL_VAE)
    eps = np.full((4, L), e)
    C = O.forward(O.MODEL_VAE, d, zero_params(O.MODEL_VAE, d), x, eps)
    assert C["kl"] == pytest.approx(kl, abs=2e-8)
    assert C["nll"] == pytest.approx(784 * math.log(2), abs=1e-9)
    if L == 2 and e == 0.0:
        assert C["loss"] == pytest.approx(543.479919, abs=1e-5)


def test_kat_vae_gmp_identical_components():
    d = O.Dims(D=784, L=64, K=10, hidden=(64,))
    x, _, _ = O.make_inputs(d, 4, O.MODEL_VAE_GMP)
    C = O.forward(O.MODEL_VAE_GMP, d, zero_params(O.MODEL_VAE_GMP, d), x, np.ones((4, 64)))
    assert C["kl"] == pytest.approx(9.41955142, abs=2e-7)


def test_param_counts():
    d = O.Dims(D=784, L=64, K=10, hidden=(64,))
    assert O.param_layout(O.MODEL_GMVAE, d)[2] == 166618
    assert O.param_layout(O.MODEL_VAE_GMP, d)[2] == 114970
    assert O.param_layout(O.MODEL_VAE, O.Dims(D=784, L=2, K=1, hidden=(64,)))[2] == 101652
    d5 = O.Dims(D=3072, L=64, K=64, hidden=(512,))
    assert O.param_layout(O.MODEL_GMVAE, d5)[2] == 4895552
    lay, P, real = O.param_layout(O.MODEL_GMVAE, d)
    assert all(off % 4 == 0 for _, _, off in lay) and P >= real


def test_flops_rule():
    d = O.Dims(D=784, L=64, K=10, hidden=(64,))
    assert O.flops_per_step(O.MODEL_GMVAE, d, 1) == pytest.approx(0.792e6, rel=2e-3)
    d5 = O.Dims(D=3072, L=64, K=64, hidden=(512,), S=50)
    assert O.flops_per_step(O.MODEL_GMVAE, d5, 1) == pytest.approx(834.7e6, rel=2e-3)


# ------------------------------------------- independent torch statement
def torch_loss(model, d, P, x, eps, u):
    """Un-simplified loss through torch.distributions (independent of oracle code)."""
    td = torch.distributions
    B, S = x.shape[0], d.S
    xf = torch.as_tensor(x, dtype=torch.float64)
    xr = xf.repeat_interleave(S, 0)
    nl = len(d.hidden) + 1

    def mlp(name, h):
        for i in range(nl if name != "prior_gmm" else 1):
            h = h @ P[f"{name}_fcnet/linear_{i}/w"] + P[f"{name}_fcnet/linear_{i}/b"]
            if i < nl - 1 and name != "prior_gmm":
                h = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "elu": torch.nn.functional.elu}[d.act](h)
        return h

    def normal(out):
        mu, raw = out[:, :d.L], out[:, d.L:]
        sig = torch.clamp_min(torch.nn.functional.softplus(raw + d.raw_sigma_bias), d.sigma_min)
        return td.Independent(td.Normal(mu, sig), 1)

    eps_t = torch.as_tensor(eps, dtype=torch.float64)
    nent = torch.zeros(B, dtype=F64)
    if model == O.MODEL_GMVAE:
        logits = mlp("encoder_y", xf)
        g = -torch.log(-torch.log(torch.as_tensor(u, dtype=torch.float64)))
        y = torch.softmax((logits.repeat_interleave(S, 0) + g) / d.temperature, -1)
        p_z = normal(mlp("prior_gmm", y))
        q_z = normal(mlp("encoder_gmm", torch.cat([xr, y], 1)))
        nent = -td.Categorical(logits=logits).entropy()
    else:
        q_z = normal(mlp("encoder", xf).repeat_interleave(S, 0))
        if model == O.MODEL_VAE:
            p_z = td.Independent(td.Normal(torch.zeros(d.L, dtype=F64), torch.ones(d.L, dtype=F64)), 1)
        else:
            p_z = td.MixtureSameFamily(
                td.Categorical(logits=P["mixture_logits"]),
                td.Independent(td.Normal(P["loc"], torch.nn.functional.softplus(P["raw_scale_diag"])), 1))
    z = q_z.base_dist.loc + q_z.base_dist.scale * eps_t
    p_x = td.Independent(td.Bernoulli(logits=mlp("decoder", z) + d.gen_bias_init), 1)
    logw = p_x.log_prob(xr) + p_z.log_prob(z) - q_z.log_prob(z) - nent.repeat_interleave(S)
    bound = torch.logsumexp(logw.reshape(B, S), 1) - math.log(S)
    return -bound.mean()


CASES = [
    ("gmvae", O.Dims(D=48, L=6, K=5, hidden=(12,)), 7),
    ("gmvae", O.Dims(D=40, L=4, K=3, hidden=(9, 7)), 5),
    ("gmvae", O.Dims(D=32, L=5, K=4, hidden=(8,), S=3, temperature=0.7, sigma_min=0.001, raw_sigma_bias=0.25), 4),
    ("vae", O.Dims(D=48, L=2, K=1, hidden=(10,)), 6),
    ("vae", O.Dims(D=30, L=3, K=1, hidden=(6,), S=4), 3),
    ("vae_gmp", O.Dims(D=48, L=6, K=5, hidden=(12,)), 7),
    ("vae_gmp", O.Dims(D=36, L=4, K=3, hidden=(8, 8), S=2, gen_bias_init=0.3), 5),
    # hidden_activation_fn other than the reference's default relu (scripts/base.py:19,90,153 accept any callable)
    ("gmvae", O.Dims(D=40, L=4, K=3, hidden=(9, 7), act="tanh"), 5),
    ("vae", O.Dims(D=48, L=2, K=1, hidden=(10,), S=2, act="sigmoid"), 6),
    ("vae_gmp", O.Dims(D=48, L=6, K=5, hidden=(12,), act="elu"), 7),
]


@pytest.mark.parametrize("name,d,B", CASES)
def test_oracle_matches_torch_autograd(name, d, B):
    model = O.MODEL_NAMES[name]
    rng = np.random.default_rng(7)
    p = O.init_params(model, d, rng)
    for k in p:                                   # non-zero biases so every path is exercised
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.1, p[k].shape)
    x, eps, u = O.make_inputs(d, B, model)
    C, g = O.loss_and_grads(model, d, p, x, eps, u)
    P = {k: torch.tensor(v, dtype=F64, requires_grad=True) for k, v in p.items()}
    loss = torch_loss(model, d, P, x, eps, u)
    loss.backward()
    assert C["loss"] == pytest.approx(loss.item(), rel=1e-12)
    for k in p:
        np.testing.assert_allclose(g[k], P[k].grad.numpy(), rtol=1e-9, atol=1e-13, err_msg=k)


@pytest.mark.parametrize("name", ["gmvae", "vae", "vae_gmp"])
def test_s1_reduction_identity(name):
    """A15: at S=1 the IWAE loss is nll + kl + nent term for term."""
    model = O.MODEL_NAMES[name]
    d = O.Dims(D=64, L=8, K=10, hidden=(16,))
    p = O.init_params(model, d, np.random.default_rng(3))
    x, eps, u = O.make_inputs(d, 16, model)
    C = O.forward(model, d, p, x, eps, u)
    assert C["loss"] == pytest.approx(C["nll"] + C["kl"] + C["nent"], rel=1e-13)


def test_fp32_restatement_within_tolerance():
    d = O.Dims(D=784, L=64, K=10, hidden=(64,))
    p = O.init_params(O.MODEL_GMVAE, d, np.random.default_rng(0))
    x, eps, u = O.make_inputs(d, 64)
    C64, g64 = O.loss_and_grads(O.MODEL_GMVAE, d, p, x, eps, u, np.float64)
    C32, g32 = O.loss_and_grads(O.MODEL_GMVAE, d, p, x, eps, u, np.float32)
    assert C32["loss"].dtype == np.float32
    assert abs(C32["loss"] - C64["loss"]) / abs(C64["loss"]) < 1e-5
    for k in g64:
        assert np.abs(g32[k] - g64[k]).max() <= 1e-4 * max(np.abs(g64[k]).max(), 1e-3), k


# ------------------------------------------------------------- TF-Adam
def test_adam_tf_epsilon_placement():
    th, m, v = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32)
    th1, _, _ = O.adam_tf_step(th, m, v, np.full(1, 1e-9, np.float32), 1)
    assert -th1[0] == pytest.approx(3.15e-6, rel=2e-2)     # torch.optim.Adam would give 9.09e-5


def test_adam_tf_three_step_trajectory():
    rng = np.random.default_rng(1)
    th = rng.normal(size=50)
    m = np.zeros(50)
    v = np.zeros(50)
    th_ref, m_ref, v_ref = th.copy(), m.copy(), v.copy()
    for t in range(1, 4):
        g = rng.normal(size=50)
        th, m, v = O.adam_tf_step(th, m, v, g, t, dtype=np.float64)
        m_ref = 0.9 * m_ref + (1 - 0.9) * g
        v_ref = 0.999 * v_ref + (1 - 0.999) * g * g
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        th_ref = th_ref - lr_t * m_ref / (np.sqrt(v_ref) + 1e-8)
    np.testing.assert_allclose(th, th_ref, rtol=1e-12)


def test_cluster_acc():
    logits = np.eye(3)[[0, 0, 0, 1, 1, 2]] * 5.0
    labels = np.array([7, 7, 3, 4, 4, 9])
    assert O.cluster_acc(logits, labels, 3) == pytest.approx(5 / 6)
    assert O.cluster_acc(logits, labels, 4) == pytest.approx(5 / 6)   # empty cluster -> mode 0
