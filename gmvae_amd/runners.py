"""Training / evaluation loops (mirror of scripts/runners.py, plumbing only).

`create_dataset`, `create_model`, `run_train`, `run_eval` keep the reference's names
(scripts/runners.py:21,65,106,235).  What is NOT reproduced: TensorBoard summaries,
image tiles, t-SNE and matplotlib plots (visualisation, out of scope -- DESIGN.md 6).
"""
from __future__ import annotations

import glob
import gzip
import os
import time
from typing import Iterator, Tuple

import numpy as np
import torch

from . import gmvae, parallel, utils, vae


# ------------------------------------------------------------------ data
def _load_mnist(root: str, split: str):
    """Local MNIST only (no network here): <root>/mnist.npz (keys x_train,y_train,x_test,y_test) or IDX files."""
    npz = os.path.join(root, "mnist.npz")
    if os.path.exists(npz):
        z = np.load(npz)
        return z[f"x_{split}"].reshape(-1, 784), z[f"y_{split}"].astype(np.int64)
    pre = "train" if split == "train" else "t10k"
    imgs = glob.glob(os.path.join(root, f"{pre}-images*"))
    labs = glob.glob(os.path.join(root, f"{pre}-labels*"))
    if imgs and labs:
        op = gzip.open if imgs[0].endswith(".gz") else open
        with op(imgs[0], "rb") as f:
            x = np.frombuffer(f.read(), np.uint8, offset=16).reshape(-1, 784)
        with op(labs[0], "rb") as f:
            y = np.frombuffer(f.read(), np.uint8, offset=8).astype(np.int64)
        return x, y
    return None


def create_dataset(config, split="train", shuffle=True, repeat=True, device="cuda"):
    """Yields (images bool/uint8 [B,784] on device, labels int64 [B]) -- the output contract of
    scripts/runners.py:21-62.  Dynamic binarisation keeps the reference's INVERTED rule
    `image < U(0,1)` (runners.py:44-47: P(pixel=1) = 1 - intensity) and is redrawn every epoch;
    examples are shuffled per epoch (the reference's batch-level shuffle order is not copied).
    The last partial batch is kept (no drop_remainder, runners.py:51).  Without local MNIST files
    (--data_dir) a synthetic Bernoulli(0.87) set with random labels stands in."""
    data = _load_mnist(getattr(config, "data_dir", "") or "", split) if getattr(config, "data_dir", None) else None
    rank, world = (parallel.dist.get_rank(), parallel.dist.get_world_size()) if parallel.dist.is_initialized() else (0, 1)
    gen = torch.Generator(device=device)
    gen.manual_seed((config.random_seed or 0) * 7919 + 17 + rank)
    if data is None:
        n = int(getattr(config, "synthetic_size", 8192))
        D = int(getattr(config, "data_dim", 784))
        inten = torch.full((n, D), 0.13, device=device)            # 1 - 0.87: P(1) = 0.87 after inversion
        labels = torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(3)).to(device)
    else:
        inten = torch.from_numpy(np.ascontiguousarray(data[0])).to(device).float() / 255.0
        labels = torch.from_numpy(data[1]).to(device)
    a, b = parallel.shard_rows(inten.shape[0], rank, world)
    inten, labels = inten[a:b], labels[a:b]
    n = inten.shape[0]
    B = int(config.batch_size)

    def it() -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        while True:
            order = torch.randperm(n, device=device, generator=gen) if shuffle else torch.arange(n, device=device)
            binar = (inten < torch.rand(inten.shape, device=device, generator=gen)).to(torch.uint8)
            for s in range(0, n, B):
                idx = order[s:s + B]
                yield binar[idx], labels[idx]
            if not repeat:
                return
    return it()


# ----------------------------------------------------------------- model
def create_model(config, data_dim):
    """scripts/runners.py:65-103: binds the flags and FIXES sigma_min=0.0, raw_sigma_bias=0.5, temperature=1.0."""
    hidden = [config.hidden_size] * config.num_layers
    ns = int(getattr(config, "n_samples", 1))
    if config.model == "gmvae":
        return gmvae.create_gmvae(data_dim, config.latent_size, mixture_components=config.mixture_components,
                                  fcnet_hidden_sizes=hidden, sigma_min=0.0, raw_sigma_bias=0.5, temperature=1.0,
                                  random_seed=config.random_seed, n_samples=ns)
    if config.model == "vae_gmp":
        return vae.create_vae(data_dim, config.latent_size, mixture_components=config.mixture_components,
                              fcnet_hidden_sizes=hidden, sigma_min=0.0, raw_sigma_bias=0.5,
                              random_seed=config.random_seed, n_samples=ns)
    if config.model == "vae":
        return vae.create_vae(data_dim, config.latent_size, fcnet_hidden_sizes=hidden, sigma_min=0.0,
                              raw_sigma_bias=0.5, random_seed=config.random_seed, n_samples=ns)
    raise ValueError(f"unknown model {config.model!r}")


def _logdir(config):
    """<logdir>/<model>/h<hidden>_n<layers>_z<latent> (scripts/runners.py:212-217)."""
    return os.path.join(config.logdir, config.model,
                        f"h{config.hidden_size}_n{config.num_layers}_z{config.latent_size}")


def _ckpt(config):
    return os.path.join(_logdir(config), "model.pt")


def run_train(config):
    """scripts/runners.py:106-232.  One iteration = the reference's sess.run([train_op, global_step])."""
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    data_dim = int(getattr(config, "data_dim", 784))
    model = create_model(config, data_dim)
    eng = model._engine
    os.makedirs(_logdir(config), exist_ok=True)
    if os.path.exists(_ckpt(config)):                      # MonitoredTrainingSession auto-restore
        model.load_state_dict(torch.load(_ckpt(config), map_location="cpu"))
        if rank == 0:
            print(f"restored {_ckpt(config)} at step {eng.global_step}")
    data = create_dataset(config, "train", shuffle=True, repeat=True)
    hook = utils.EarlyStoppingHook(config.early_stop_rounds, config.early_stop_threshold)
    last_save, t0, s0 = time.time(), time.time(), eng.global_step
    losses = []
    while eng.global_step <= config.max_steps:              # `<=`: the reference runs one extra step (runners.py:231)
        images, labels = next(data)
        tail = eng.train_step(images, lr=config.learning_rate)        # fwd + bwd + all-reduce + Adam
        losses.append(tail[0] / tail[4])                    # device tensor: no per-step host sync
        if eng.global_step % config.summarise_every == 0 or eng.global_step > config.max_steps:
            vals = torch.stack(losses).tolist()
            base = eng.global_step - len(vals)
            losses = []
            stop = False
            for i, v in enumerate(vals):                    # EarlyStoppingHook sees every step's (all-reduced) loss
                stop = hook.after_run(base + i + 1, v) or stop
            if rank == 0:
                rate = (eng.global_step - s0) / max(time.time() - t0, 1e-9)
                msg = f"Step {eng.global_step}, loss: {vals[-1]:f}  ({rate:.1f} global_step/sec)"
                if config.model == "gmvae":
                    q = model.encoder_y(images).distribution.logits
                    msg += f"  cluster_acc {utils.cluster_acc(q, labels, config.mixture_components).item():.4f}"
                print(msg, flush=True)
            if stop:
                if rank == 0:
                    print("[Early Stopping Criterion Satisfied]")
                break
        if rank == 0 and time.time() - last_save > 120:     # save_checkpoint_secs=120 (runners.py:226)
            torch.save(model.state_dict(), _ckpt(config))
            last_save = time.time()
    if rank == 0:
        torch.save(model.state_dict(), _ckpt(config))
    return model


@torch.no_grad()
def run_eval(config):
    """scripts/runners.py:235-459 without the plots.  Reports the true per-example loss and, under a
    different name, the reference's figure (sum of per-batch MEANS / number of examples,
    runners.py:298,330-335 -- i.e. roughly loss / batch_size)."""
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    model = create_model(config, int(getattr(config, "data_dim", 784)))
    if not os.path.exists(_ckpt(config)):
        raise FileNotFoundError(f"no checkpoint at {_ckpt(config)} (the reference would poll every 60 s)")
    model.load_state_dict(torch.load(_ckpt(config), map_location="cpu"))
    eng = model._engine
    tot = torch.zeros(5, device=eng.device)
    ref_sum, n_batches, codes, ys = 0.0, 0, [], []
    for images, labels in create_dataset(config, config.split, shuffle=False, repeat=False):
        o = eng.forward(images)
        tot += o["tail"][:5]
        ref_sum += (o["tail"][0] / o["tail"][4]).item()
        n_batches += 1
        codes.append(o["z"])
        if o["y"] is not None:
            ys.append(o["y"])
    if world > 1:
        parallel.dist.all_reduce(tot)
    n = tot[4].item()
    res = {f"{config.split}/loss_per_example": tot[0].item() / n, f"{config.split}/nll": tot[1].item() / n,
           f"{config.split}/kl_div_z": tot[2].item() / n, f"{config.split}/nent": tot[3].item() / n,
           f"{config.split}/reference_misnormalised_loss_per_example": ref_sum / n, "examples": int(n)}
    if rank == 0:
        for k, v in res.items():
            print(f"{k}: {v}")
    return res
