"""the input pipeline's dynamic binarisation (scripts/runners.py:44-47) and the
data-parallel definition of cluster_acc (scripts/utils.py:173-191).  CPU part: the oracle's Philox generator against
the published Random123 known-answer vectors, the binarisation's statistics, the histogram form of cluster_acc.
GPU part (-m gpu): the HIP kernel against the oracle, bit for bit."""
import numpy as np
import pytest
import torch

import oracle as O


def test_philox_known_answer_vectors():
    # Random123 (Salmon et al., SC'11) kat_vectors: philox4x32 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = O.philox4x32_10(np.array([ctr], dtype=np.uint32), key[0], key[1])[0]
        assert tuple(int(v) for v in got) == want


def test_binarize_oracle_statistics_and_determinism():
    rng = np.random.default_rng(0)
    pix = np.repeat(np.arange(256, dtype=np.uint8)[None, :], 4096, axis=0)        # column j has intensity j
    pix = np.concatenate([pix, rng.integers(0, 256, (4096, 4), dtype=np.uint8)], axis=1)[:, :260]
    x = O.binarize(pix, np.arange(4096), seed=5, step=0)
    assert x.dtype == np.uint8 and set(np.unique(x)) <= {0, 1}
    p1 = x[:, :256].mean(axis=0)                                                  # P[x = 1] = 1 - intensity / 255
    assert np.abs(p1 - (1 - np.arange(256) / 255.0)).max() < 0.04
    assert x[:, 255].sum() == 0 and x[:, 0].mean() > 0.999                        # 255/255 < u never; 0 < u almost surely
    assert np.array_equal(x, O.binarize(pix, np.arange(4096), seed=5, step=0))    # same (seed, step): same bits
    assert (x != O.binarize(pix, np.arange(4096), seed=5, step=1)).mean() > 0.2   # a new step is a new draw
    rows = rng.permutation(4096)[:100]
    assert O.binarize(pix, rows, 5, 0).shape == (100, 260)


def test_cluster_acc_histogram_form_matches_reference_form():
    rng = np.random.default_rng(1)
    for K in (3, 10, 64):
        logits = rng.normal(size=(777, K))
        labels = rng.integers(0, 10, 777)
        hist = np.zeros((K, 10), dtype=np.int64)
        np.add.at(hist, (logits.argmax(1), labels), 1)
        assert abs(O.cluster_acc_from_hist(hist) - O.cluster_acc(logits, labels, K)) < 1e-12
        from gmvae_amd.utils import cluster_acc_from_hist
        assert abs(float(cluster_acc_from_hist(torch.from_numpy(hist))) - O.cluster_acc(logits, labels, K)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("B,D", [(1024, 784), (100, 784), (7, 12), (4096, 3072)])
def test_hip_binarize_matches_oracle_bit_for_bit(B, D):
    from gmvae_amd.data import binarize
    rng = np.random.default_rng(B + D)
    N = B + 37
    pix = rng.integers(0, 256, (N, D), dtype=np.uint8)
    pt = torch.from_numpy(pix).cuda()
    rows = rng.permutation(N)[:B].astype(np.int32)
    for seed, step in ((0, 0), (0x1234567890abcdef, 3), (7, 2 ** 40 + 5)):
        got = binarize(pt, rows=torch.from_numpy(rows).cuda(), seed=seed, step=step).cpu().numpy()
        assert np.array_equal(got, O.binarize(pix, rows, seed, step))
    got = binarize(pt, row0=5, batch=B, seed=9, step=1).cpu().numpy()                # contiguous rows, no index
    assert np.array_equal(got, O.binarize(pix, np.arange(5, 5 + B), 9, 1))
    # a data-parallel shard (global row offset in the Philox counter): the shard of the whole = the shard drawn alone
    whole = binarize(pt, rows=torch.from_numpy(rows).cuda(), seed=3, step=2)
    h = B // 2
    part = binarize(pt, rows=torch.from_numpy(rows[h:]).cuda(), seed=3, step=2, out_row0=h)
    assert torch.equal(part, whole[h:])
    assert np.array_equal(part.cpu().numpy(), O.binarize(pix, rows[h:], 3, 2, out_row0=h))


@pytest.mark.gpu
def test_device_dataset_epochs_and_training_step():
    from gmvae_amd.data import DeviceDataset
    from gmvae_amd.engine import Engine
    rng = np.random.default_rng(3)
    N, B = 1000, 256
    pix = rng.integers(0, 256, (N, 28, 28, 1), dtype=np.uint8)
    lab = rng.integers(0, 10, N)
    ds = DeviceDataset(pix, lab, shuffle=True, seed=11)
    seen = torch.cat([ds.next_rows(B) for _ in range(4)])[:N]                        # one epoch = every row once
    assert sorted(seen.cpu().tolist()) == list(range(N))
    x, y = ds.next_batch(B)
    assert x.shape == (B, 784) and x.dtype == torch.uint8 and int(x.max()) <= 1 and y.shape == (B,)
    e = Engine("gmvae", 784, 16, 10, [64], random_seed=0)
    e.train_step(x, lr=1e-3)
    l0 = float(e.grads[e.P]) / B
    for _ in range(30):
        xb, _ = ds.next_batch(B)
        e.train_step(xb, lr=1e-3)
    l1 = float(e.grads[e.P]) / B
    assert np.isfinite(l1) and l1 < l0                                                # it trains on the device pipeline


@pytest.mark.gpu
def test_pipeline_graph_equals_binarize_then_step(monkeypatch):
    """The train graph that starts from raw pixels == gmvae_binarize + the training step, step by step (same rows,
    same (seed, step) uniforms; the separate first-layer launch in every step makes the two paths the same kernels)."""
    monkeypatch.setenv("GMVAE_NO_FL", "1")
    from gmvae_amd.data import DeviceDataset, binarize
    from gmvae_amd.engine import Engine
    rng = np.random.default_rng(8)
    N, B, n = 3000, 1024, 3
    pix = rng.integers(0, 256, (N, 784), dtype=np.uint8)
    e1 = Engine("gmvae", 784, 64, 10, [64], random_seed=4)
    ds1 = DeviceDataset(pix, shuffle=True, seed=21)
    replay = e1.capture_train_pipeline(ds1, B, lr=1e-3, n_steps=n)
    replay()
    replay()
    torch.cuda.synchronize()
    e2 = Engine("gmvae", 784, 64, 10, [64], random_seed=4)
    ds2 = DeviceDataset(pix, shuffle=True, seed=21)
    for step in range(2 * n):
        rows = ds2.next_rows(B)
        x = binarize(ds2.pixels, rows=rows, seed=e2.noise_seed ^ Engine.BINARIZE_SEED_XOR, step=step)
        if step == 2 * n - 1:
            assert torch.equal(x, replay.batches[n - 1]) and torch.equal(rows, replay.rows[n - 1])
        e2.train_step(x, lr=1e-3)
    torch.cuda.synchronize()
    assert e1.global_step == e2.global_step == 2 * n
    assert torch.equal(e1.params, e2.params)
