"""-m gpu: the parts of the reference's construction surface that round 2 refused or ignored (VERDICT r2 "missing" 2-6),
each against the oracle.

* vector `bias_init` of ConditionalBernoulli (scripts/base.py:102-103,135): ABI v3 `GmvaeDims.gen_bias_vec`;
* the learned mixture prior at ANY K, L (scripts/vae.py:231-244): the tiled log-prob where the LDS-resident one does not fit;
* `initializers` (scripts/base.py:18,49-50) applied at bind time;
* run_eval's prior draws (scripts/runners.py:274-292)."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import build_hip
    build_hip.build(verbose=False)
    from tests import hip_util
    return hip_util


def _params(model, d, seed):
    rng = np.random.default_rng(seed)
    p = O.init_params(model, d, rng)
    for k in p:
        if k.endswith("/b"):
            p[k] = rng.normal(0, 0.05, p[k].shape)
    return p


VEC_CASES = [
    ("gmvae", O.Dims(D=784, L=64, K=10, hidden=(64,)), 256),      # default sizes: the chain schedule (decoder layer = GEMM launch)
    ("gmvae", O.Dims(D=784, L=128, K=10, hidden=(512,)), 64),     # bin/run_train.sh sizes: general schedule, interior tiles
    ("vae", O.Dims(D=200, L=8, K=1, hidden=(32,), S=3), 19),      # ragged tiles, IWAE rows
    ("vae_gmp", O.Dims(D=97, L=6, K=5, hidden=(24, 24)), 33),     # D % 4 != 0: the scalar epilogue path
]


@pytest.mark.parametrize("name,d,B", VEC_CASES, ids=[f"{n}-D{d.D}-B{B}" for n, d, B in VEC_CASES])
def test_vector_bias_init_step_matches_oracle(H, name, d, B):
    """loss and every gradient with logits = MLP(z) + vector (scripts/base.py:135), e.g. the logit of the data mean."""
    model = O.MODEL_NAMES[name]
    vec = np.random.default_rng(7).normal(0.5, 1.0, d.D).astype(np.float32)
    dv = O.Dims(**{**d.__dict__, "gen_bias_init": vec})
    p = _params(model, d, B)
    x, eps, u = O.make_inputs(d, B, model)
    loss_v, _ = H.compare_step(model, dv, p, x, eps, u, grad_rtol=2e-4 if d.S > 1 else 1e-4)
    loss_0, _ = H.compare_step(model, d, p, x, eps, u, grad_rtol=2e-4 if d.S > 1 else 1e-4)
    assert abs(loss_v - loss_0) > 1.0                              # the vector took effect


def test_vector_bias_init_through_the_model_api():
    """create_gmvae(gen_bias_init=<vector>): run_model, decoder(z) and the train graph all see it."""
    import gmvae_amd
    D, Lz, K, B = 784, 64, 10, 128
    vec = torch.linspace(-2.0, 2.0, D)
    m = gmvae_amd.create_gmvae(D, Lz, mixture_components=K, fcnet_hidden_sizes=[64], sigma_min=0.0, raw_sigma_bias=0.5,
                               gen_bias_init=vec, random_seed=5)
    d = O.Dims(D=D, L=Lz, K=K, hidden=(64,), gen_bias_init=vec.numpy())
    x, eps, u = O.make_inputs(d, B)
    xt = torch.from_numpy(x).cuda()
    loss = m.run_model(xt, xt, None, eps=torch.from_numpy(eps), u=torch.from_numpy(u))
    flat = m._engine.params.detach().cpu().numpy().astype(np.float64)
    Cc = O.forward(O.MODEL_GMVAE, d, O.unpack(O.MODEL_GMVAE, d, flat), x, eps, u, np.float64)
    assert abs(loss.item() - Cc["loss"]) <= 1e-4 * abs(Cc["loss"])
    z = torch.from_numpy(Cc["z"].astype(np.float32)).cuda()
    lam = m.decoder(z).logits.cpu().numpy()
    np.testing.assert_allclose(lam, Cc["lam"], rtol=0, atol=2e-4 * np.abs(Cc["lam"]).max())
    # three graph steps == three eager steps (the graph must not take a schedule that ignores the vector)
    e = m._engine
    xs = torch.from_numpy((np.random.default_rng(1).random((3, B, D)) < 0.87).astype(np.uint8)).cuda()
    p0, seed = e.params.detach().clone(), e.noise_seed
    sx, replay = e.capture_train_step(B, n_steps=3)
    sx.copy_(xs)
    replay()
    pg = e.params.detach().clone()
    with torch.no_grad():
        e.params.copy_(p0); e.m.zero_(); e.v.zero_()
    e.global_step = 0
    e.step_dev.zero_()
    for i in range(3):
        e.train_step(xs[i])
    torch.cuda.synchronize()
    assert (pg - e.params.detach()).abs().max().item() < 2e-5 and e.noise_seed == seed
    with pytest.raises(ValueError):
        gmvae_amd.create_gmvae(D, Lz, mixture_components=K, fcnet_hidden_sizes=[64], gen_bias_init=torch.zeros(5))


GMP_CASES = [
    (O.Dims(D=128, L=24, K=100, hidden=(32,)), 50, {}),                    # K > 64
    (O.Dims(D=64, L=300, K=40, hidden=(48,)), 21, {}),                     # (loc, 1/s) image past the LDS-resident form
    (O.Dims(D=96, L=70, K=130, hidden=(40, 24), S=2), 17, {}),             # both, ragged everywhere, IWAE rows
    (O.Dims(D=784, L=64, K=10, hidden=(64,)), 96, {"GMVAE_GMP_TILED": "1", "GMVAE_NO_MEGA": "1"}),   # tiled form forced
]


@pytest.mark.parametrize("d,B,env", GMP_CASES, ids=[f"K{d.K}-L{d.L}-B{B}" for d, B, _ in GMP_CASES])
def test_mixture_prior_of_any_size(H, monkeypatch, d, B, env):
    """MixtureSameFamily.log_prob + its gradients (scripts/vae.py:231-244) where K > 64 or K L exceeds LDS."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model = O.MODEL_VAE_GMP
    p = _params(model, d, B)
    p["loc"] = np.random.default_rng(3).normal(0, 1.0, p["loc"].shape)          # separated components: resp not uniform
    x, eps, u = O.make_inputs(d, B, model)
    H.compare_step(model, d, p, x, eps, u, grad_rtol=2e-4 if d.S > 1 else 1e-4)
    tail, rows, z, _, _ = H.hip_forward(model, d, O.pack(model, d, p, np.float32), x, eps, None)
    Cc = O.forward(model, d, {k: np.asarray(v, np.float32).astype(np.float64) for k, v in p.items()}, x, eps, None, np.float64)
    np.testing.assert_allclose(rows[:, 2], Cc["logp"], rtol=0, atol=1e-4 * np.abs(Cc["logp"]).max())


def test_custom_initializers_are_applied_at_bind_time():
    """scripts/base.py:18,49-50: `initializers` maps 'w' / 'b' to initializers; here callables shape -> array."""
    import gmvae_amd
    from gmvae_amd import base, _lib as L
    m = gmvae_amd.create_gmvae(100, 8, mixture_components=5, fcnet_hidden_sizes=[16], random_seed=1)
    e = m._engine
    before = {k: v.clone() for k, v in e.views().items()}
    dec = base.ConditionalBernoulli(size=100, hidden_layer_sizes=[16], name="decoder",
                                    initializers={"w": lambda s: np.full(s, 0.25, np.float32), "b": lambda s: np.ones(s)})
    dec.bind(e, L.NET_DECODER)
    after = e.views()
    for k in after:
        if k.startswith("decoder_fcnet/"):
            assert torch.all(after[k] == (0.25 if k.endswith("/w") else 1.0)), k
        else:
            assert torch.equal(after[k], before[k]), k
    z = torch.zeros(3, 8, device="cuda")
    assert torch.allclose(dec(z).logits, torch.full((3, 100), 1.0 + 16 * 0.25 * 1.0, device="cuda"))   # relu(b0 = 1) * 0.25 * 16 + 1
    with pytest.raises(ValueError):
        base.ConditionalNormal(size=8, hidden_layer_sizes=[16], name="no_such_net", initializers={"w": lambda s: np.zeros(s)}).bind(e, L.NET_ENCODER_GMM)
