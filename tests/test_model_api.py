"""-m gpu: the reference-shaped Python surface (create_vae / create_gmvae / run_model /
encoder* / decoder / prior*) on top of the C ABI."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64)


def test_gmvae_run_model_backward_and_adam_three_steps():
    import gmvae_amd
    d = O.Dims(D=784, L=16, K=10, hidden=(64,))
    model = gmvae_amd.create_gmvae(784, 16, mixture_components=10, fcnet_hidden_sizes=[64], sigma_min=0.0,
                                   raw_sigma_bias=0.5, random_seed=3)
    e = model._engine
    assert e.P_real == O.param_layout(O.MODEL_GMVAE, d)[2]
    flat = _np(e.params).astype(np.float32)
    m = np.zeros_like(flat)
    v = np.zeros_like(flat)
    for t in range(1, 4):
        x, eps, u = O.make_inputs(d, 64, seed_x=t, seed_noise=10 + t)
        xt = torch.from_numpy(x).cuda()
        model.params.grad = None
        loss = model.run_model(xt, xt, None, eps=torch.from_numpy(eps), u=torch.from_numpy(u))
        loss.backward()
        flat, m, v, C, g = O.train_step(O.MODEL_GMVAE, d, flat.astype(np.float64), m.astype(np.float64),
                                        v.astype(np.float64), t, x, eps, u, dtype=np.float64)
        assert loss.item() == pytest.approx(C["loss"], rel=1e-4)
        np.testing.assert_allclose(_np(model.params.grad), g, atol=1e-4 * np.abs(g).max())
        s = model.summaries
        assert s["elbo"].item() == pytest.approx(-C["loss"], rel=1e-4)
        assert s["nent"].item() == pytest.approx(C["nent"], abs=1e-4)
        e.adam(1e-3)
        assert e.global_step == t
    # dead-ReLU / tiny-gradient coordinates make Adam's m/(sqrt(v)+eps) ill-conditioned: compare where |g| is sane
    np.testing.assert_allclose(_np(e.params), flat, atol=2e-4)
    assert np.abs(_np(e.params) - flat).mean() < 2e-6


def test_vae_gmp_modules_and_aux_methods():
    import gmvae_amd
    model = gmvae_amd.create_vae(784, 8, mixture_components=5, fcnet_hidden_sizes=[32], sigma_min=0.0,
                                 raw_sigma_bias=0.5, random_seed=1)
    d = O.Dims(D=784, L=8, K=5, hidden=(32,))
    p = O.unpack(O.MODEL_VAE_GMP, d, _np(model.params))
    x, eps, _ = O.make_inputs(d, 12, O.MODEL_VAE_GMP)
    xt = torch.from_numpy(x).cuda()
    q = model.encoder(xt)
    C = O.forward(O.MODEL_VAE_GMP, d, p, x, eps)
    np.testing.assert_allclose(_np(q.mean()), C["mu_q"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(_np(q.scale_diag), C["sig_q"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(_np(model.transform(xt)), C["mu_q"], rtol=1e-4, atol=1e-5)
    zt = torch.from_numpy(C["z"]).float().cuda()
    np.testing.assert_allclose(_np(model.decoder(zt).log_prob(xt)), C["logpx"], rtol=1e-5)
    np.testing.assert_allclose(_np(model.prior().log_prob(zt)), C["logp"], rtol=1e-4)
    assert model.reconstruct_images(xt).shape == (12, 784)
    assert model.generate_samples(7).shape == (7, 8)
    assert model.generate_sample_images(num_samples=3).shape == (3, 784)
    loss = model.run_model(xt, xt, eps=torch.from_numpy(eps))
    assert loss.item() == pytest.approx(C["loss"], rel=1e-4)
    assert model.mix_components == 5


def test_gmvae_modules_match_oracle():
    import gmvae_amd
    model = gmvae_amd.create_gmvae(200, 6, mixture_components=4, fcnet_hidden_sizes=[24, 16], sigma_min=0.0,
                                   raw_sigma_bias=0.5, temperature=0.8, random_seed=2)
    d = O.Dims(D=200, L=6, K=4, hidden=(24, 16), temperature=0.8)
    p = O.unpack(O.MODEL_GMVAE, d, _np(model.params))
    x, eps, u = O.make_inputs(d, 10)
    C = O.forward(O.MODEL_GMVAE, d, p, x, eps, u)
    xt = torch.from_numpy(x).cuda()
    q_y = model.encoder_y(xt)
    np.testing.assert_allclose(_np(q_y.distribution.logits), C["logits"], rtol=1e-4, atol=1e-5)
    y = q_y.sample(uniform=torch.from_numpy(u).cuda())
    np.testing.assert_allclose(_np(y), C["y"], rtol=1e-4, atol=1e-6)
    pz = model.prior_gmm(y)
    np.testing.assert_allclose(_np(pz.loc), C["mu_p"], rtol=1e-4, atol=1e-5)
    qz = model.encoder_gmm(xt, y)
    np.testing.assert_allclose(_np(qz.scale_diag), C["sig_q"], rtol=1e-4, atol=1e-5)
    assert model.generate_samples(3).shape == (12, 6)
    assert model.generate_samples(2, clusters=[0, 3, 3]).shape == (6, 6)
    assert model.transform(xt).shape == (10, 6)
    labels = torch.randint(0, 10, (10,))
    loss = model.run_model(xt, xt, labels, eps=torch.from_numpy(eps), u=torch.from_numpy(u))
    assert loss.item() == pytest.approx(C["loss"], rel=1e-4)
    acc = model.summaries["cluster_acc"].item()
    # the accuracy is invariant to the mode tie-break, so the device value is the reference's to fp32 rounding
    assert acc == pytest.approx(O.cluster_acc(_np(q_y.distribution.logits), labels.numpy(), 4), abs=1e-6)


@pytest.mark.parametrize("fn,act", [(torch.tanh, "tanh"), (torch.sigmoid, "sigmoid"), (torch.nn.functional.elu, "elu")])
def test_gmvae_with_another_hidden_activation_matches_oracle(fn, act):
    """hidden_activation_fn (scripts/gmvae.py:282 passes ONE callable to every conditional's MLP; scripts/base.py:19,90,153):
    the accessors (gmvae_mlp_forward) and run_model + backward on tanh / sigmoid / ELU against the oracle."""
    import gmvae_amd
    model = gmvae_amd.create_gmvae(200, 6, mixture_components=4, fcnet_hidden_sizes=[24, 16], sigma_min=0.0, raw_sigma_bias=0.5,
                                   temperature=0.8, random_seed=2, hidden_activation_fn=fn)
    d = O.Dims(D=200, L=6, K=4, hidden=(24, 16), temperature=0.8, act=act)
    p = O.unpack(O.MODEL_GMVAE, d, _np(model.params))
    x, eps, u = O.make_inputs(d, 10)
    C, g = O.loss_and_grads(O.MODEL_GMVAE, d, p, x, eps, u, np.float64)
    xt = torch.from_numpy(x).cuda()
    q_y = model.encoder_y(xt)
    np.testing.assert_allclose(_np(q_y.distribution.logits), C["logits"], rtol=1e-4, atol=1e-5)
    y = q_y.sample(uniform=torch.from_numpy(u).cuda())
    qz = model.encoder_gmm(xt, y)
    np.testing.assert_allclose(_np(qz.scale_diag), C["sig_q"], rtol=1e-4, atol=1e-5)
    loss = model.run_model(xt, xt, None, eps=torch.from_numpy(eps), u=torch.from_numpy(u))
    assert loss.item() == pytest.approx(C["loss"], rel=1e-4)
    loss.backward()
    gref = O.pack(O.MODEL_GMVAE, d, g, np.float64)
    got = _np(model._engine.params.grad).astype(np.float64)
    assert np.abs(got - gref).max() <= 1e-4 * np.abs(gref).max()


def _torch_draws(seed, u_shape=None, eps_shape=None):
    """The draws base.RelaxedOneHotCategorical.sample / MultivariateNormalDiag.sample make for `seed` (each call seeds
    a fresh device generator, as the reference passes the same `seed=self.random_seed` to every sampler)."""
    out = []
    if u_shape is not None:
        g = torch.Generator(device="cuda"); g.manual_seed(seed)
        out.append(torch.rand(u_shape, device="cuda", generator=g).clamp_min(1.1754943508222875e-38).cpu().numpy())
    if eps_shape is not None:
        g = torch.Generator(device="cuda"); g.manual_seed(seed)
        out.append(torch.randn(eps_shape, device="cuda", generator=g).cpu().numpy())
    return out


def test_gmvae_eval_methods_values_match_oracle():
    """scripts/gmvae.py:109-188 by VALUE: reconstruct_images, transform (a SAMPLED code for the GMVAE),
    generate_samples(clusters) and generate_sample_images, with the samplers' draws injected into the oracle."""
    import gmvae_amd
    from oracle import gmvae_oracle as OO
    seed, B = 7, 10
    model = gmvae_amd.create_gmvae(200, 6, mixture_components=4, fcnet_hidden_sizes=[24], sigma_min=0.0,
                                   raw_sigma_bias=0.5, temperature=0.8, random_seed=seed)
    d = O.Dims(D=200, L=6, K=4, hidden=(24,), temperature=0.8)
    p = O.unpack(O.MODEL_GMVAE, d, _np(model.params))
    x, _, _ = O.make_inputs(d, B)
    xt = torch.from_numpy(x).cuda()
    u, eps = _torch_draws(seed, (B, 4), (B, 6))
    C = O.forward(O.MODEL_GMVAE, d, p, x, eps, u)
    np.testing.assert_allclose(_np(model.transform(xt)), C["z"], rtol=1e-4, atol=1e-5)                     # gmvae.py:140-149
    np.testing.assert_allclose(_np(model.reconstruct_images(xt)), OO.sigmoid(C["lam"]), rtol=1e-4, atol=1e-5)   # :109-123
    # generate_samples(num_samples, clusters): z = mu_p(y_k) + sigma_p(y_k) * eps, rows ordered [sample][cluster]
    clusters, n = [0, 3, 3], 5
    y = np.eye(4)[clusters]
    pp = y @ p["prior_gmm_fcnet/linear_0/w"] + p["prior_gmm_fcnet/linear_0/b"]
    mu_p, sig_p, _ = OO._normal_head(pp, 6, np.float64(0.5), np.float64(0.0))
    (eps_g,) = _torch_draws(seed, None, (n, 3, 6))
    want = (mu_p[None] + sig_p[None] * eps_g).reshape(n * 3, 6)
    got = model.generate_samples(n, clusters=clusters)
    np.testing.assert_allclose(_np(got), want, rtol=1e-4, atol=1e-5)                                        # gmvae.py:152-188
    lam, _ = OO._mlp_fwd({k: np.asarray(v, np.float64) for k, v in p.items()}, "decoder", 2, want)
    np.testing.assert_allclose(_np(model.generate_sample_images(got)), OO.sigmoid(lam), rtol=1e-4, atol=1e-5)
    # all K clusters when none are given: K * num_samples rows
    (eps_a,) = _torch_draws(seed, None, (2, 4, 6))
    pp = np.eye(4) @ p["prior_gmm_fcnet/linear_0/w"] + p["prior_gmm_fcnet/linear_0/b"]
    mu_a, sig_a, _ = OO._normal_head(pp, 6, np.float64(0.5), np.float64(0.0))
    np.testing.assert_allclose(_np(model.generate_samples(2)), (mu_a[None] + sig_a[None] * eps_a).reshape(8, 6), rtol=1e-4, atol=1e-5)


def test_vae_eval_methods_values_match_oracle():
    """scripts/vae.py:80-123: transform returns the MEAN code; reconstruct_images decodes a SAMPLED code."""
    import gmvae_amd
    from oracle import gmvae_oracle as OO
    seed, B = 5, 9
    model = gmvae_amd.create_vae(784, 8, fcnet_hidden_sizes=[32], sigma_min=0.0, raw_sigma_bias=0.5, random_seed=seed)
    d = O.Dims(D=784, L=8, K=1, hidden=(32,))
    p = O.unpack(O.MODEL_VAE, d, _np(model.params))
    x, _, _ = O.make_inputs(d, B, O.MODEL_VAE)
    xt = torch.from_numpy(x).cuda()
    (eps,) = _torch_draws(seed, None, (B, 8))
    C = O.forward(O.MODEL_VAE, d, p, x, eps)
    np.testing.assert_allclose(_np(model.transform(xt)), C["mu_q"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(_np(model.reconstruct_images(xt)), OO.sigmoid(C["lam"]), rtol=1e-4, atol=1e-5)
    (eps_s,) = _torch_draws(seed, None, (4, 8))
    np.testing.assert_allclose(_np(model.generate_samples(4)), eps_s, rtol=1e-5, atol=1e-6)           # N(0, I) prior


def test_state_dict_roundtrip_uses_tf_variable_names():
    import gmvae_amd
    m1 = gmvae_amd.create_gmvae(64, 4, mixture_components=3, fcnet_hidden_sizes=[8], random_seed=1)
    sd = m1.state_dict()
    assert "encoder_gmm_fcnet/linear_0/w" in sd and sd["encoder_gmm_fcnet/linear_0/w"].shape == (67, 8)
    assert "prior_gmm_fcnet/linear_0/b" in sd and "global_step" in sd
    m2 = gmvae_amd.create_gmvae(64, 4, mixture_components=3, fcnet_hidden_sizes=[8], random_seed=2)
    assert not torch.equal(m1.params, m2.params)
    m2.load_state_dict(sd)
    assert torch.equal(m1.params.detach(), m2.params.detach())


def test_graph_replay_matches_eager_steps():
    import gmvae_amd
    from gmvae_amd.engine import Engine
    x, _, _ = O.make_inputs(O.Dims(D=784, L=16, K=10, hidden=(64,)), 128)
    xt = torch.from_numpy(x).cuda()
    outs = []
    for graph in (False, True):
        e = Engine("gmvae", 784, 16, 10, [64], random_seed=5)
        if graph:
            sx, replay = e.capture_train_step(128, lr=1e-3)
            sx.copy_(xt)
            for _ in range(3):
                replay()
        else:
            for _ in range(3):
                e.train_step(xt, lr=1e-3)
        torch.cuda.synchronize()
        outs.append((e.params.detach().clone(), e.global_step))
    assert outs[0][1] == outs[1][1] == 3
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-6)


@pytest.mark.parametrize("in_launch_first_layer", [False, True])
def test_multi_step_graph_matches_eager_steps_on_the_same_batches(in_launch_first_layer, monkeypatch):
    """n_steps batches in ONE graph launch == the same batches stepped one by one: same Philox (seed, step) stream,
    same order.  With the separate first-layer launch in every step (GMVAE_NO_FL) the kernels are the same and the
    result is bit-identical; by default steps 2..n run the first layer inside mega_fwd_bwd, on weight images the
    previous finalize_adam scattered and with noise drawn in the kernel -- same values, another summation order."""
    from gmvae_amd.engine import Engine
    if not in_launch_first_layer:
        monkeypatch.setenv("GMVAE_NO_FL", "1")
    n, B = 4, 1024
    rng = np.random.default_rng(3)
    xs = torch.from_numpy((rng.random((n, B, 784)) < 0.87).astype(np.uint8)).cuda()
    outs = []
    for multi in (False, True):
        e = Engine("gmvae", 784, 64, 10, [64], random_seed=5)
        if multi:
            sx, replay = e.capture_train_step(B, lr=1e-3, n_steps=n)
            assert tuple(sx.shape) == (n, B, 784)
            sx.copy_(xs)
            replay()
            replay()
        else:
            for r in range(2):
                for i in range(n):
                    e.train_step(xs[i], lr=1e-3)
        torch.cuda.synchronize()
        outs.append((e.params.detach().clone(), e.global_step, e.grads[e.P:].clone()))
    assert outs[0][1] == outs[1][1] == 2 * n
    if not in_launch_first_layer:
        assert torch.equal(outs[0][0], outs[1][0])
        assert torch.equal(outs[0][2], outs[1][2])
    else:
        assert torch.isfinite(outs[1][0]).all()
        # 8 Adam steps of lr 1e-3 move a weight by <= 8e-3; the two paths may differ by rounding in the gradients only
        _assert_same_up_to_gradient_rounding(outs[0][0], outs[1][0])
        assert abs(outs[0][2][0].item() - outs[1][2][0].item()) < 1e-4 * abs(outs[0][2][0].item())


def _assert_same_up_to_gradient_rounding(p_a, p_b):
    """Parameters after a few TF-Adam steps on two launch schedules that differ by rounding in the gradients only: equal
    to 2e-5 except where |g| of the first steps is of the order of Adam's epsilon (the update lr * g / (|g| + 1e-8) then
    turns a 1e-9 gradient difference into a 5e-5 step: a handful of first-layer weights of nearly-always-off pixels)."""
    d = (p_a - p_b).abs()
    assert d.max().item() < 2e-4
    assert (d > 2e-5).float().mean().item() < 1e-4
    assert d.mean().item() < 1e-7



@pytest.mark.parametrize("model,L_,K_,B", [("vae", 8, 1, 256), ("gmvae", 16, 10, 100), ("gmvae", 64, 10, 1000),
                                          ("vae_gmp", 64, 10, 256), ("vae", 2, 1, 100)])
def test_in_launch_first_layer_other_models_and_sizes(model, L_, K_, B):
    """Steps 2..n of a train graph run the first layer inside mega_fwd_bwd also for the VAE (one first-layer tensor,
    4 column tiles), for sizes other than the specialised instance's and for a ragged last panel (B = 100, 1000);
    they must track eager steps on the same batches and leave the hand-off error word clear."""
    from gmvae_amd.engine import Engine
    n = 3
    rng = np.random.default_rng(11)
    xs = torch.from_numpy((rng.random((n, B, 784)) < 0.87).astype(np.uint8)).cuda()
    outs = []
    for multi in (False, True):
        e = Engine(model, 784, L_, K_, [64], random_seed=2)
        if multi:
            sx, replay = e.capture_train_step(B, lr=1e-3, n_steps=n)
            sx.copy_(xs)
            replay()
            replay()
            assert e.handoff_timeouts() == 0
        else:
            for r in range(2):
                for i in range(n):
                    e.train_step(xs[i], lr=1e-3)
        torch.cuda.synchronize()
        outs.append((e.params.detach().clone(), e.grads[e.P:].clone()))
    assert torch.isfinite(outs[1][0]).all()
    _assert_same_up_to_gradient_rounding(outs[0][0], outs[1][0])
    assert abs(outs[0][1][0].item() - outs[1][1][0].item()) < 1e-4 * abs(outs[0][1][0].item())


def test_handoff_error_word_fails_fast_and_recovers(monkeypatch):
    """A set hand-off error word (what a timed-out wait leaves behind) makes every later wait give up at its first
    failed poll (the step is then correct or NaN-poisoned, never stalled, never stale: granules are only accepted
    with this step's tag); the word stays set until drop_graphs() clears it, after which the schedule without mutual
    waits (GMVAE_NO_FL) runs clean."""
    import ctypes as C
    import time
    from gmvae_amd import _lib as L
    from gmvae_amd.engine import Engine
    B, n = 1024, 3
    rng = np.random.default_rng(1)
    xs = torch.from_numpy((rng.random((n, B, 784)) < 0.87).astype(np.uint8)).cuda()
    e = Engine("gmvae", 784, 64, 10, [64], random_seed=1)
    sx, replay = e.capture_train_step(B, lr=1e-3, n_steps=n)
    sx.copy_(xs)
    replay()
    torch.cuda.synchronize()
    assert e.handoff_timeouts() == 0 and np.isfinite(e.grads[e.P].item())
    good = e.params.detach().clone()
    d, ws = e._workspace(B)
    off = C.c_uint64()
    L.check(L.lib.gmvae_workspace_offset(C.byref(d), e.model, b"sync", C.byref(off)), "off")
    ws.view(torch.int32)[off.value // 4 + 1] = 1                     # as if a wait had timed out
    t0 = time.perf_counter()
    replay()
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 1.0                            # no stall
    assert e.handoff_timeouts() == 1
    monkeypatch.setenv("GMVAE_NO_FL", "1")
    e.drop_graphs()
    with torch.no_grad():
        e.params.copy_(good)
        e.m.zero_()
        e.v.zero_()
    sx, replay = e.capture_train_step(B, lr=1e-3, n_steps=n)
    sx.copy_(xs)
    replay()
    torch.cuda.synchronize()
    assert e.handoff_timeouts() == 0 and np.isfinite(e.grads[e.P].item()) and torch.isfinite(e.params).all()


def test_missing_engine_and_bad_activation_fail_loudly():
    import gmvae_amd
    from gmvae_amd import base
    with pytest.raises(NotImplementedError):
        gmvae_amd.create_gmvae(64, 4, 3, hidden_activation_fn=torch.sin)
    m = gmvae_amd.TrainableGMVAE(3, None, None, None, None)
    with pytest.raises(RuntimeError):
        m.run_model(torch.zeros(1, 1), torch.zeros(1, 1), None)
    with pytest.raises(RuntimeError):
        base.ConditionalNormal(4, [8]).condition([torch.zeros(1, 4)])
