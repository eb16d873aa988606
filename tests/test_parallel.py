"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: row sharding, the single flat
all-reduce with the loss sums and count in the tail, and the replicated TF-Adam update.  The
per-rank compute is the oracle (no GPU here); the -m gpu twin below runs the HIP step on one GPU
split into two 'ranks' in-process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O
from gmvae_amd import parallel as par


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_buffer(model, d, flat, x, eps, u):
    """What gmvae_step leaves in the flat buffer: gradient SUMS + [loss, nll, kl, nent, count] sums."""
    B = x.shape[0]
    C, g = O.loss_and_grads(model, d, O.unpack(model, d, flat), x, eps, u, np.float64)
    P = flat.size
    buf = np.zeros(P + par.TAIL)
    buf[:P] = O.pack(model, d, g, np.float64) * B
    buf[P:P + 5] = [C["loss"] * B, C["nll"] * B, C["kl"] * B, C["nent"] * B, B]
    return buf


def _worker(rank, world, port, Bg, steps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = par.init_from_env("gloo")
    assert (r, w) == (rank, world)
    model, d = O.MODEL_GMVAE, O.Dims(D=40, L=4, K=3, hidden=(8,))
    flat = O.pack(model, d, O.init_params(model, d, np.random.default_rng(0)), np.float64)
    m, v = np.zeros_like(flat), np.zeros_like(flat)
    P = flat.size
    losses = []
    for t in range(1, steps + 1):
        x, eps, u = O.make_inputs(d, Bg, seed_x=t, seed_noise=100 + t)
        a, b = par.shard_rows(Bg, rank, world)
        buf = torch.from_numpy(_local_buffer(model, d, flat, x[a:b], eps[a:b], u[a:b]))
        par.all_reduce_flat(buf)                       # the ONE collective of the step
        scale = par.grad_scale(buf, P).item()
        assert buf[P + 4].item() == Bg
        flat, m, v = O.adam_tf_step(flat, m, v, buf[:P].numpy() * scale, t, dtype=np.float64)
        losses.append(buf[P].item() * scale)
    assert par.assert_replicas_identical(torch.from_numpy(flat))
    out[rank] = (flat, losses)
    dist.destroy_process_group()


@pytest.mark.parametrize("Bg", [16, 15])            # 15: uneven shards (8 + 7 rows)
def test_two_rank_gloo_equals_single_process(Bg):
    steps, world = 3, 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), Bg, steps, out), nprocs=world, join=True)
    model, d = O.MODEL_GMVAE, O.Dims(D=40, L=4, K=3, hidden=(8,))
    flat = O.pack(model, d, O.init_params(model, d, np.random.default_rng(0)), np.float64)
    m, v = np.zeros_like(flat), np.zeros_like(flat)
    ref_losses = []
    for t in range(1, steps + 1):
        x, eps, u = O.make_inputs(d, Bg, seed_x=t, seed_noise=100 + t)
        flat, m, v, C, _ = O.train_step(model, d, flat, m, v, t, x, eps, u, dtype=np.float64)
        ref_losses.append(C["loss"])
    f0, l0 = out[0]
    f1, l1 = out[1]
    assert np.array_equal(f0, f1)                                   # replicas bit-identical
    np.testing.assert_allclose(l0, ref_losses, rtol=1e-12)
    # Adam divides by sqrt(v)+eps: reduction-order noise (1e-16) on near-dead coordinates is amplified
    np.testing.assert_allclose(f0, flat, rtol=0, atol=1e-9)


def _hist_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    par.init_from_env("gloo")
    from gmvae_amd.utils import cluster_acc_from_hist
    rng = np.random.default_rng(5)
    logits, labels = rng.normal(size=(300, 10)), rng.integers(0, 10, 300)
    a, b = par.shard_rows(300, rank, world)
    hist = torch.zeros(10, 10, dtype=torch.int64)                      # what cluster_hist leaves on each rank
    hist.index_put_((torch.from_numpy(logits[a:b].argmax(1)), torch.from_numpy(labels[a:b])), torch.ones(b - a, dtype=torch.int64),
                    accumulate=True)
    dist.all_reduce(hist)                                              # the collective of utils.cluster_acc(all_reduce=True)
    out[rank] = float(cluster_acc_from_hist(hist))
    dist.destroy_process_group()


def test_two_rank_cluster_acc_is_the_global_batch_accuracy():
    """the data-parallel definition of cluster_acc -- sum the [K, n_labels] histograms over the
    ranks, then take each cluster's majority -- equals scripts/utils.py:173-191 on the unsharded batch."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_hist_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    rng = np.random.default_rng(5)
    logits, labels = rng.normal(size=(300, 10)), rng.integers(0, 10, 300)
    want = O.cluster_acc(logits, labels, 10)
    assert abs(out[0] - want) < 1e-6 and out[0] == out[1]


def test_shard_rows_cover_and_balance():
    for n in (1, 7, 16, 1000, 1023):
        for w in (1, 2, 3, 8):
            spans = [par.shard_rows(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
def test_hip_step_sharded_sum_equals_full_batch():
    """1-GPU stand-in for the N-GPU equality: gradient SUMS of two row shards add up to the full-batch
    buffer (what the all-reduce computes), to fp32 reduction-order tolerance."""
    import hip_util as H
    d = O.Dims(D=784, L=16, K=10, hidden=(64,))
    model = O.MODEL_GMVAE
    flat = O.pack(model, d, O.init_params(model, d, np.random.default_rng(1)), np.float32)
    x, eps, u = O.make_inputs(d, 256)
    full, tail = H.hip_step(model, d, flat, x, eps, u)
    acc, tacc = 0.0, 0.0
    for r in range(2):
        a, b = par.shard_rows(256, r, 2)
        g, t = H.hip_step(model, d, flat, x[a:b], eps[a:b], u[a:b])
        acc, tacc = acc + g, tacc + t
    assert tacc[4] == tail[4] == 256
    assert abs(tacc[0] - tail[0]) <= 1e-5 * abs(tail[0])
    assert np.abs(acc - full).max() <= 2e-5 * np.abs(full).max()


@pytest.mark.gpu
def test_rccl_in_library_world1_matches_plain_step():
    """The C-side RCCL path (communicator of one rank on this GPU): gmvae_dp_step and the captured DP graph
    give the same trajectory as step + Adam without any collective."""
    from gmvae_amd.engine import Engine
    x, _, _ = O.make_inputs(O.Dims(D=784, L=16, K=10, hidden=(64,)), 128)
    xt = torch.from_numpy(x).cuda()
    ref = Engine("gmvae", 784, 16, 10, [64], random_seed=5)
    for _ in range(4):
        ref.train_step(xt, lr=1e-3)
    e = Engine("gmvae", 784, 16, 10, [64], random_seed=5)
    e.enable_rccl()
    e.dp_step(xt, 1e-3)
    e.dp_step(xt, 1e-3)
    sx, replay = e.capture_train_step(128, lr=1e-3, all_reduce=True)
    sx.copy_(xt)
    replay()
    replay()
    torch.cuda.synchronize()
    assert e.global_step == ref.global_step == 4
    assert torch.allclose(e.params.detach(), ref.params.detach(), atol=1e-6)
    assert int(e.step_dev[0].item()) == 4


@pytest.mark.gpu
def test_rccl_multi_step_graph_world1_tracks_eager_steps():
    """The data-parallel train graph with several steps per launch: from its second step on the first layer runs inside
    mega_fwd_bwd on weight images that the Adam launch AFTER the all-reduce scattered (adam_tf_img).  With a
    one-rank communicator the all-reduce is the identity, so the trajectory must track plain eager steps."""
    from gmvae_amd.engine import Engine
    rng = np.random.default_rng(12)
    n, B = 3, 1024
    xs = torch.from_numpy((rng.random((n, B, 784)) < 0.87).astype(np.uint8)).cuda()
    ref = Engine("gmvae", 784, 64, 10, [64], random_seed=6)
    for r in range(2):
        for i in range(n):
            ref.train_step(xs[i], lr=1e-3)
    e = Engine("gmvae", 784, 64, 10, [64], random_seed=6)
    e.enable_rccl()
    sx, replay = e.capture_train_step(B, lr=1e-3, all_reduce=True, n_steps=n)
    assert e.dp_mode == "rccl-in-hipgraph" and tuple(sx.shape) == (n, B, 784)
    sx.copy_(xs)
    replay()
    replay()
    torch.cuda.synchronize()
    assert e.global_step == ref.global_step == 2 * n and int(e.step_dev[0].item()) == 2 * n
    assert e.handoff_timeouts() == 0 and torch.isfinite(e.params).all()
    assert (e.params - ref.params).abs().max().item() < 2e-5
    assert abs(e.grads[e.P].item() - ref.grads[ref.P].item()) < 1e-4 * abs(ref.grads[ref.P].item())
