"""Flag table parity (CPU) and a short end-to-end train/eval run (GPU)."""
import pytest

from gmvae_amd import run_gmvae, utils

REFERENCE_FLAGS = {   # scripts/run_gmvae.py:11-58
    "mode": "train", "model": "gmvae", "latent_size": 8, "hidden_size": 64, "num_layers": 1,
    "mixture_components": 10, "batch_size": 16, "logdir": "/tmp/smc_vi", "random_seed": None,
    "learning_rate": 0.001, "max_steps": int(1e9), "early_stop_rounds": 1000, "early_stop_threshold": 0.001,
    "summarise_every": 50, "gpu_id": "0", "gpu_num": "0", "num_samples": 10, "num_generations": 10, "split": "train"}


def test_flag_names_and_defaults_match_reference():
    cfg = vars(run_gmvae.build_parser().parse_args([]))
    for k, v in REFERENCE_FLAGS.items():
        assert cfg[k] == v, k


def test_early_stopping_hook_semantics():
    h = utils.EarlyStoppingHook(max_steps=3, threshold=0.1)
    assert h.after_run(1, 100.0) is False            # first call only records the step (utils.py:38-45)
    assert h.after_run(2, 100.0) is False            # sets prev_loss
    assert h.after_run(3, 95.0) is False             # not a 10 % improvement: counter 1
    assert h.after_run(4, 80.0) is False             # improvement: reset
    assert [h.after_run(s, 79.0) for s in (5, 6, 7)] == [False, False, True]
    assert h.after_run(2, 1.0) is False              # global step went backwards -> reset (recovery)


def test_wait_for_checkpoint_polls_like_the_reference(tmp_path):
    """scripts/utils.py:100-111: loop + sleep until the checkpoint appears (no GPU needed)."""
    import threading
    import time
    from gmvae_amd import runners
    path = tmp_path / "model.pt"
    with pytest.raises(FileNotFoundError):
        runners.wait_for_checkpoint(str(path), poll_seconds=0.01, max_wait=0.05)
    threading.Timer(0.15, lambda: path.write_bytes(b"x")).start()
    t0 = time.time()
    assert runners.wait_for_checkpoint(str(path), poll_seconds=0.02, max_wait=5.0) == str(path)
    assert 0.1 < time.time() - t0 < 3.0
    assert runners._graph_steps(50) == 25 and runners._graph_steps(20) == 20 and runners._graph_steps(7) == 7
    assert runners._graph_steps(1000) == 25 and runners._graph_steps(97) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["gmvae", "vae_gmp"])
def test_run_train_takes_the_graph_path_and_equals_the_eager_loop(tmp_path, model):
    """The drop-in loop (scripts/runners.py:222-232) runs summarise_every-aligned pipeline graphs: binarisation, noise,
    step and Adam on the device.  Same batches through eager launches (config.eager) -> the same trajectory."""
    import torch
    from gmvae_amd import runners
    outs = []
    for eager in (False, True):
        args = ["--mode=train", f"--model={model}", "--latent_size=64", "--batch_size=1024", "--max_steps=44",
                "--summarise_every=15", f"--logdir={tmp_path}/{eager}", "--random_seed=3", "--synthetic_size=5000"]
        m = run_gmvae.main(args + (["--eager"] if eager else []))
        assert runners.run_train.last_path == ("eager" if eager else "pipeline-graph")
        assert m._engine.global_step == 45 and m._engine.handoff_timeouts() == 0
        outs.append((m._engine.params.detach().clone(), m._engine.grads[m._engine.P:].clone()))
    assert torch.isfinite(outs[0][0]).all()
    # 45 Adam steps; the schedules differ in summation order only.  Adam's m / sqrt(v) amplifies that on coordinates
    # with near-zero gradients (the learned prior's variables), hence a mean gate, a loose max gate and the last loss.
    diff = (outs[0][0] - outs[1][0]).abs()
    assert diff.mean().item() < 5e-6 and diff.max().item() < 45 * 1e-3 * 0.05
    assert abs(outs[0][1][0].item() - outs[1][1][0].item()) < 2e-5 * abs(outs[1][1][0].item())


@pytest.mark.gpu
def test_restart_equivalence_bit_for_bit(tmp_path):
    """k steps -> checkpoint -> NEW model object restored from it -> k steps == 2k uninterrupted steps, bit for bit
    (parameters, both Adam slots, step counter, Philox stream position).  MonitoredTrainingSession's restore,
    scripts/runners.py:222-228; checkpoint keys are the TF variable names."""
    import numpy as np
    import torch
    import gmvae_amd
    k, B = 5, 256
    xs = torch.from_numpy((np.random.default_rng(0).random((2 * k, B, 784)) < 0.87).astype(np.uint8)).cuda()
    mk = lambda: gmvae_amd.create_gmvae(784, 16, mixture_components=10, fcnet_hidden_sizes=[64], sigma_min=0.0,
                                        raw_sigma_bias=0.5, random_seed=9)
    a = mk()
    for i in range(2 * k):
        a._engine.train_step(xs[i], lr=1e-3)
    b = mk()
    for i in range(k):
        b._engine.train_step(xs[i], lr=1e-3)
    path = tmp_path / "model.pt"
    torch.save(b.state_dict(), path)
    sd = torch.load(path, map_location="cpu")
    # TF checkpoint names: the variable, its two Adam slots, the optimizer's power accumulators, the step
    for key in ("encoder_gmm_fcnet/linear_0/w", "encoder_gmm_fcnet/linear_0/w/Adam", "encoder_gmm_fcnet/linear_0/w/Adam_1",
                "decoder_fcnet/linear_1/b/Adam", "beta1_power", "beta2_power", "global_step"):
        assert key in sd, key
    assert sd["encoder_gmm_fcnet/linear_0/w/Adam"].shape == sd["encoder_gmm_fcnet/linear_0/w"].shape == (794, 64)
    assert float(sd["beta1_power"]) == pytest.approx(0.9 ** (k + 1)) and int(sd["global_step"]) == k
    c = gmvae_amd.create_gmvae(784, 16, mixture_components=10, fcnet_hidden_sizes=[64], sigma_min=0.0, raw_sigma_bias=0.5,
                               random_seed=1234)       # another init, another noise seed: everything must come from the file
    c.load_state_dict(sd)
    assert c._engine.global_step == k and int(c._engine.step_dev[0].item()) == k
    # the second half through a multi-step train graph: the graph path restarts from the file too
    sx, replay = c._engine.capture_train_step(B, lr=1e-3, n_steps=k)
    sx.copy_(xs[k:])
    d = mk()
    d.load_state_dict(sd)
    for i in range(k, 2 * k):
        d._engine.train_step(xs[i], lr=1e-3)
    replay()
    torch.cuda.synchronize()
    for got in (d,):
        assert torch.equal(got._engine.params, a._engine.params)
        assert torch.equal(got._engine.m, a._engine.m) and torch.equal(got._engine.v, a._engine.v)
        assert got._engine.global_step == 2 * k
    assert c._engine.global_step == 2 * k and (c._engine.params - a._engine.params).abs().max().item() < 2e-5


@pytest.mark.gpu
def test_train_then_eval_end_to_end(tmp_path):
    args = ["--model=gmvae", "--latent_size=16", "--batch_size=256", "--max_steps=60", "--summarise_every=20",
            f"--logdir={tmp_path}", "--random_seed=1", "--synthetic_size=2048"]
    model = run_gmvae.main(["--mode=train"] + args)
    assert model._engine.global_step == 61              # `<=` runs one extra step, as the reference does
    from gmvae_amd import runners
    assert runners.run_train.last_path == "pipeline-graph"
    res = run_gmvae.main(["--mode=eval"] + args)
    assert res["examples"] == 2048
    assert res["train/loss_per_example"] < 450          # Bernoulli(0.87) data: well below the D ln 2 = 543 start
    assert res["train/reference_misnormalised_loss_per_example"] == pytest.approx(
        res["train/loss_per_example"] / 256, rel=0.05)
    # the tensors the reference's eval graph also produces (scripts/runners.py:274-292): latent state + labels over the
    # split, num_samples prior draws ([num_samples * K, L] for the GMVAE), decoded draws stacked with ONE component's
    import torch
    assert res["latent_state"].shape == (2048, 16) and res["labels"].shape == (2048,)
    assert res["samples"].shape == (10 * 10, 16) and torch.isfinite(res["samples"]).all()
    assert res["sample_images"].shape == (2, 10 * 10, 28, 28, 1) and 0 <= res["sampled_cluster"] < 10
    assert float(res["sample_images"].min()) >= 0.0 and float(res["sample_images"].max()) <= 1.0      # Bernoulli means


@pytest.mark.gpu
def test_run_train_degrades_to_the_safe_schedule_instead_of_dying(tmp_path, monkeypatch):
    """A hand-off timeout of the fused schedule (fault hook: the error word a timed-out wait leaves behind) must not end
    training (MonitoredTrainingSession recovers, scripts/runners.py:222-232): run_train drops its
    graphs, re-captures on the schedule without mutual waits, goes on from the last good step and logs once.  A second
    fault on the safe schedule is an error."""
    import torch
    from gmvae_amd import runners
    for k in ("GMVAE_NO_FL", "GMVAE_MEGA_Q"):
        monkeypatch.delenv(k, raising=False)
    calls = []

    def fault_once(eng):
        calls.append(eng.global_step)
        if len(calls) == 1:
            eng.inject_handoff_fault()

    base = ["--mode=train", "--model=gmvae", "--latent_size=64", "--batch_size=1024", "--max_steps=59", "--summarise_every=20",
            "--random_seed=2", "--synthetic_size=4096"]
    cfg = run_gmvae.build_parser().parse_args(base + [f"--logdir={tmp_path}/a"])
    cfg.fault_hook = fault_once
    try:
        m = runners.run_train(cfg)
        e = m._engine
        assert runners.run_train.degraded and e.safe_schedule and e.handoff_timeouts() == 0
        import os
        assert "GMVAE_NO_FL" not in os.environ and "GMVAE_MEGA_Q" not in os.environ     # per engine (GmvaeDims.sched_flags), not per process
        assert e.dims(1024).sched_flags == 1
        assert e.global_step == 60 and torch.isfinite(e.params).all() and (tmp_path / "a/gmvae/h64_n1_z64/model.pt").exists()
        assert calls[0] == 20 and len(calls) >= 3
        last = (e.grads[e.P] / e.grads[e.P + 4]).item()
        assert 0 < last < 500                                # it kept training: D ln 2 - ln K = 541 at the start
        # a clean run of the same seed in the SAME process takes the fast schedule again (nothing process-wide was
        # switched) and ends close by (the faulted block's steps were either complete or skipped and re-run on other batches)
        cfg2 = run_gmvae.build_parser().parse_args(base + [f"--logdir={tmp_path}/b"])
        m2 = runners.run_train(cfg2)
        assert not runners.run_train.degraded and not m2._engine.safe_schedule
        last2 = (m2._engine.grads[m2._engine.P] / m2._engine.grads[m2._engine.P + 4]).item()
        assert abs(last - last2) < 0.05 * abs(last2)
        # a fault that persists on the safe schedule raises
        cfg3 = run_gmvae.build_parser().parse_args(base + [f"--logdir={tmp_path}/c"])
        cfg3.fault_hook = lambda eng: eng.inject_handoff_fault()
        with pytest.raises(RuntimeError, match="without mutual waits"):
            runners.run_train(cfg3)
        # a non-finite loss with NO hand-off timeout is divergence, not a schedule problem: it raises as such and does
        # not switch schedules (it used to be reported as a hand-off timeout and re-run on the slow schedule)
        cfg4 = run_gmvae.build_parser().parse_args(base + [f"--logdir={tmp_path}/d"])
        seen = []

        def poison_params(eng):
            seen.append(eng)
            if len(seen) == 1:
                eng.params.detach().fill_(float("nan"))
        cfg4.fault_hook = poison_params
        with pytest.raises(RuntimeError, match="no hand-off timeout"):
            runners.run_train(cfg4)
        assert not runners.run_train.degraded and not seen[0].safe_schedule
    finally:
        pass


@pytest.mark.gpu
def test_run_train_at_the_reference_shipped_sizes(tmp_path):
    """bin/run_train.sh:3-14 (latent 128, hidden 512, batch 64): the loop runs pipeline graphs of the skinny schedule
    (csrc/skinny.hpp) and equals the eager loop on the same batches."""
    import torch
    from gmvae_amd import runners
    outs = []
    for eager in (False, True):
        args = ["--mode=train", "--model=gmvae", "--latent_size=128", "--hidden_size=512", "--batch_size=64", "--max_steps=29",
                "--summarise_every=10", f"--logdir={tmp_path}/{eager}", "--random_seed=4", "--synthetic_size=2048"]
        m = run_gmvae.main(args + (["--eager"] if eager else []))
        assert runners.run_train.last_path == ("eager" if eager else "pipeline-graph") and m._engine.global_step == 30
        outs.append((m._engine.params.detach().clone(), (m._engine.grads[m._engine.P] / m._engine.grads[m._engine.P + 4]).item()))
    assert torch.isfinite(outs[0][0]).all() and 0 < outs[0][1] < 560
    # same launches up to the optimizer: the graph's W launch applies TF-Adam itself (alpha_t from expm1f), the eager loop
    # calls adam_tf_step (alpha_t from fp64 pow): ~3e-7 relative on the step size, 30 steps
    diff = (outs[0][0] - outs[1][0]).abs()
    assert diff.mean().item() < 1e-6 and diff.max().item() < 30 * 1e-3 * 0.05
    assert abs(outs[0][1] - outs[1][1]) < 2e-5 * abs(outs[1][1])


FLAG_SWEEP = [   # model, latent, hidden, layers, batch: the schedules a reference user's flag choices land on
    ("gmvae", 8, 64, 1, 16),        # scripts/run_gmvae.py defaults: mega schedule, one ragged panel
    ("gmvae", 8, 128, 1, 32),       # latent 8 with a wider layer: general schedule
    ("gmvae", 32, 256, 1, 48),      # skinny schedule, ragged row tiles
    ("gmvae", 16, 64, 2, 40),       # two hidden layers: general schedule
    ("vae", 2, 64, 1, 100),         # BASELINE configs[0]
    ("vae", 16, 256, 2, 24),
    ("vae_gmp", 64, 64, 1, 256),    # BASELINE configs[1]
    ("vae_gmp", 24, 128, 1, 30),    # general schedule with the learned mixture prior
]


@pytest.mark.gpu
@pytest.mark.parametrize("model,latent,hidden,layers,batch", FLAG_SWEEP, ids=[f"{m}-z{l}-h{h}x{n}-b{b}" for m, l, h, n, b in FLAG_SWEEP])
def test_train_then_eval_over_the_flag_space(tmp_path, model, latent, hidden, layers, batch):
    """`run_gmvae --mode=train` then `--mode=eval` for flag combinations that land on every schedule (mega, skinny, chain,
    general; one and two hidden layers; all three models): the loss falls from its start, the checkpoint round-trips, the
    evaluation's per-example loss is finite and below the untrained value."""
    import math
    args = [f"--model={model}", f"--latent_size={latent}", f"--hidden_size={hidden}", f"--num_layers={layers}",
            f"--batch_size={batch}", "--max_steps=59", "--summarise_every=20", f"--logdir={tmp_path}", "--random_seed=5",
            "--synthetic_size=1024", "--mixture_components=10"]
    m = run_gmvae.main(["--mode=train"] + args)
    e = m._engine
    assert e.global_step == 60 and e.handoff_timeouts() == 0
    last = (e.grads[e.P] / e.grads[e.P + 4]).item()
    start = 784 * math.log(2.0)                      # ~ the untrained loss (GMVAE: minus ln K)
    assert math.isfinite(last) and last < 0.9 * start
    res = run_gmvae.main(["--mode=eval"] + args)
    assert res["examples"] == 1024 and math.isfinite(res["train/loss_per_example"]) and res["train/loss_per_example"] < 0.9 * start
    assert res["latent_state"].shape == (1024, latent)


@pytest.mark.gpu
@pytest.mark.parametrize("model,latent,batch", [("gmvae", 64, 1024), ("vae_gmp", 64, 256), ("gmvae", 128, 64)])
def test_verify_every_replays_launches_through_the_two_launch_form(tmp_path, monkeypatch, model, latent, batch):
    """GMVAE_VERIFY_EVERY=n (gmvae_amd/runners.py _verify_launch): every n-th train-graph launch is replayed from a snapshot on a
    shadow engine through the two-launch form and compared bit for bit -- the canary for the one-launch steps' plain loads behind
    flags (csrc/mega3.hpp) on firmware / partitions other than the one they were verified on.  Here: it runs, it agrees (one-launch
    GMVAE and VAE_GMP steps; a skinny configuration, where both forms are the same kernels), and a corrupted trajectory is caught."""
    from gmvae_amd import runners
    monkeypatch.setenv("GMVAE_VERIFY_EVERY", "2")
    hidden = 64 if latent == 64 else 512
    args = ["--mode=train", f"--model={model}", f"--latent_size={latent}", f"--hidden_size={hidden}", f"--batch_size={batch}",
            "--max_steps=59", "--summarise_every=15", f"--logdir={tmp_path}/a", "--random_seed=3", "--synthetic_size=5000"]
    m = run_gmvae.main(args)
    assert runners.run_train.last_path == "pipeline-graph" and runners.run_train.launches == 4 and runners.run_train.verified_launches == 2
    assert m._engine.global_step == 60 and m._engine.handoff_timeouts() == 0
    # a trajectory that does NOT match its replay (the snapshot is taken, then one parameter is nudged behind the launch's back)
    real = runners._verify_launch

    def nudged(eng, snap, batches, g, lr):
        with torch.no_grad():
            eng.params[5] += 1e-3
        return real(eng, snap, batches, g, lr)

    import torch
    monkeypatch.setattr(runners, "_verify_launch", nudged)
    with pytest.raises(RuntimeError, match="GMVAE_VERIFY_EVERY"):
        run_gmvae.main([a_ if not a_.startswith("--logdir") else f"--logdir={tmp_path}/b" for a_ in args])
