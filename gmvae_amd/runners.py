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


def select_device(config, local_rank: int) -> int:
    """The device of this process.  Under a launcher (LOCAL_RANK set: one process per GPU) it is the local rank.
    Otherwise the reference's flags decide (scripts/run_gmvae.py:45-48, scripts/runners.py:193,209): `--gpu_num` is the
    comma-separated list of visible physical devices (gpu_options.visible_device_list) and `--gpu_id` indexes INTO that
    list (tf.device('/gpu:<gpu_id>'))."""
    if "LOCAL_RANK" in os.environ:
        return int(local_rank)
    visible = [int(t) for t in str(getattr(config, "gpu_num", "0") or "0").split(",") if t.strip() != ""] or [0]
    idx = int(getattr(config, "gpu_id", "0") or 0)
    if not 0 <= idx < len(visible):
        raise ValueError(f"--gpu_id={idx} does not index the visible device list --gpu_num={visible}")
    dev = visible[idx]
    if not 0 <= dev < torch.cuda.device_count():
        raise ValueError(f"--gpu_num names device {dev}, but this host has {torch.cuda.device_count()} GPU(s)")
    return dev


def _logdir(config):
    """<logdir>/<model>/h<hidden>_n<layers>_z<latent> (scripts/runners.py:212-217)."""
    return os.path.join(config.logdir, config.model,
                        f"h{config.hidden_size}_n{config.num_layers}_z{config.latent_size}")


def _ckpt(config):
    return os.path.join(_logdir(config), "model.pt")


def _save_checkpoint(model, path: str):
    """Atomic: a reader (run_eval's wait_for_checkpoint, a restart) never sees a half-written file."""
    tmp = path + ".tmp"
    torch.save(model.state_dict(), tmp)
    os.replace(tmp, path)


def wait_for_checkpoint(path: str, poll_seconds: float = 60.0, max_wait: float | None = None) -> str:
    """scripts/utils.py:100-111: loop until a checkpoint exists, sleeping `poll_seconds` (60 s in the reference)
    between looks.  max_wait (build-side addition, None = forever like the reference) bounds the wait."""
    t0 = time.time()
    while not os.path.exists(path):
        if max_wait is not None and time.time() - t0 >= max_wait:
            raise FileNotFoundError(f"no checkpoint at {path} after waiting {max_wait:.0f} s")
        print(f"Checkpoint not found in {os.path.dirname(path)}, sleeping for {poll_seconds:g} seconds.", flush=True)
        time.sleep(poll_seconds)
    return path


def create_device_dataset(config, split="train", shuffle=True):
    """The input pipeline of scripts/runners.py:21-62 with the raw uint8 pixels resident in HBM (data.DeviceDataset):
    this rank's contiguous shard of the split's rows (+ labels), shuffled per epoch on the device; binarisation happens
    per step on the device (gmvae_binarize: inside the train graph, or as a launch in the eager loop).  Without local
    MNIST files a synthetic set of intensity 33/255 (P[x = 1] = 0.87 after the reference's inverted rule) stands in."""
    from .data import DeviceDataset
    data = _load_mnist(getattr(config, "data_dir", "") or "", split) if getattr(config, "data_dir", None) else None
    rank, world = (parallel.dist.get_rank(), parallel.dist.get_world_size()) if parallel.dist.is_initialized() else (0, 1)
    if data is None:
        n = int(getattr(config, "synthetic_size", 8192))
        D = int(getattr(config, "data_dim", 784))
        pix = np.full((n, D), 33, dtype=np.uint8)
        lab = np.random.default_rng(3).integers(0, 10, n)
    else:
        pix, lab = np.ascontiguousarray(data[0]), data[1]
    a, b = parallel.shard_rows(pix.shape[0], rank, world)
    return DeviceDataset(pix[a:b], lab[a:b], shuffle=shuffle, seed=(config.random_seed or 0) * 7919 + 17 + rank)


def _graph_steps(every: int, cap: int = 32) -> int:
    """Steps per graph launch: the largest divisor of summarise_every up to `cap`, so that summaries fall on launch
    boundaries (one host sync per summary, none in between)."""
    return max(g for g in range(1, max(1, min(cap, every)) + 1) if every % g == 0)


def _verify_launch(eng, snap, batches, g, lr):
    """GMVAE_VERIFY_EVERY=n (debug): the g steps a train-graph launch just took, replayed from a snapshot of (params, m, v,
    step) on a SHADOW engine through the two-launch form (GMVAE_NO_FUSE=1: mega2_fwd_bwd -> dw_adam, no plain loads behind flags
    inside a launch) on the same binarised batches and the same noise keys, and compared BIT FOR BIT.  The one-launch steps
    (csrc/mega3.hpp) are correct under the cache behaviour they were verified on (gfx950, ROCm 7.2: INTEGRATION.md); a stale
    line would give silently wrong gradients, not a timeout -- this is the canary for other firmware / partitions."""
    from .engine import Engine
    p0, m0, v0, step0 = snap
    hp = dict(eng.hp)
    gb = eng.gen_bias_vec if eng.gen_bias_vec is not None else hp.pop("gen_bias_init")
    hp.pop("gen_bias_init", None)
    sh = Engine(eng.model_name, eng.D, eng.Lz, eng.K, eng.hidden, n_samples=eng.S, gen_bias_init=gb, random_seed=0, **hp)
    sh.rank, sh.noise_seed = eng.rank, eng.noise_seed
    with torch.no_grad():
        sh.params.copy_(p0); sh.m.copy_(m0); sh.v.copy_(v0)
    sh.global_step = step0
    sh.step_dev.fill_(step0)
    old = os.environ.get("GMVAE_NO_FUSE")
    os.environ["GMVAE_NO_FUSE"] = "1"
    try:
        sx, rp = sh.capture_train_step(batches.shape[-2], lr=lr, n_steps=g)
        sx.copy_(batches if g > 1 else batches.reshape(sx.shape))
        rp()
        torch.cuda.synchronize()
    finally:
        if old is None:
            del os.environ["GMVAE_NO_FUSE"]
        else:
            os.environ["GMVAE_NO_FUSE"] = old
        sh.drop_graphs()
    bad = [nm for nm, a_, b_ in (("params", sh.params, eng.params), ("Adam m", sh.m, eng.m), ("Adam v", sh.v, eng.v))
           if not torch.equal(a_.detach(), b_.detach())]
    if bad:
        d = (sh.params.detach() - eng.params.detach()).abs()
        raise RuntimeError(f"GMVAE_VERIFY_EVERY: steps {step0 + 1}..{step0 + g} of the train graph differ from the two-launch form in "
                           f"{bad} (max |d params| {d.max().item():.3e}, {int((d > 0).sum().item())} elements): the one-launch "
                           f"step's plain loads saw stale data on this device -- run with GMVAE_NO_FUSE=1")


def run_train(config):
    """scripts/runners.py:106-232.  One iteration = the reference's sess.run([train_op, global_step]); here
    `summarise_every`-aligned hipGraph launches of several steps each: binarisation from the resident pixels, Philox
    noise, forward, backward, (RCCL all-reduce,) TF-Adam and the per-step loss log all run on the device, and the host
    looks only at summary steps: the logging hook's line, the early-stopping hook fed with EVERY step's (all-reduced)
    loss from the log (scripts/utils.py:13-57), the checkpoint timer (save_checkpoint_secs=120, runners.py:226).
    `config.eager = True` (or GMVAE_EAGER_TRAIN=1) runs the same batches through eager launches instead."""
    from .data import binarize
    from .engine import Engine
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(select_device(config, local))
    data_dim = int(getattr(config, "data_dim", 784))
    model = create_model(config, data_dim)
    eng = model._engine
    os.makedirs(_logdir(config), exist_ok=True)
    if os.path.exists(_ckpt(config)):                      # MonitoredTrainingSession auto-restore
        model.load_state_dict(torch.load(_ckpt(config), map_location="cpu"))
        if rank == 0:
            print(f"restored {_ckpt(config)} at step {eng.global_step}")
    eng.sync_replicas()            # default --random_seed=None: every process drew its own init; rank 0's wins
    ds = create_device_dataset(config, "train", shuffle=True)
    B, lr, every = int(config.batch_size), float(config.learning_rate), int(config.summarise_every)
    eager = bool(getattr(config, "eager", False)) or os.environ.get("GMVAE_EAGER_TRAIN") == "1"
    bseed = eng.noise_seed ^ Engine.BINARIZE_SEED_XOR
    G = _graph_steps(every)
    dp_graph = False
    if world > 1 and not eager and parallel.dist.get_backend() == "nccl":     # (in-library RCCL: one device per rank)
        try:
            eng.enable_rccl()                               # raises on EVERY rank if it fails on any
            dp_graph = True
        except Exception as e:
            if rank == 0:
                print(f"[run_train] in-library RCCL unavailable ({e}); torch.distributed all-reduce", flush=True)
    run_train.last_path = "eager" if eager else ("dp-graph" if world > 1 else "pipeline-graph")
    verify_every = int(os.environ.get("GMVAE_VERIFY_EVERY", "0") or 0)      # debug canary: see _verify_launch
    run_train.launches = run_train.verified_launches = 0
    run_train.degraded = False
    hook = utils.EarlyStoppingHook(config.early_stop_rounds, config.early_stop_threshold)
    last_save, t0, s0 = time.time(), time.time(), eng.global_step
    logs = []                                               # device tensors [n, TAIL]: no host sync until a summary
    last_x = last_rows = None

    def run(n):
        """n training steps on the next n batches of the pipeline."""
        nonlocal last_x, last_rows
        while n > 0:
            g = G if n >= G else 1
            if eager:
                rows = ds.next_rows(B)
                x = binarize(ds.pixels, rows=rows, seed=bseed, step=eng.global_step, out_row0=rank * B)
                logs.append(eng.train_step(x, lr=lr).clone().view(1, -1))
                last_x, last_rows, g = x, rows, 1
            elif world == 1:
                replay = eng.capture_train_pipeline(ds, B, lr=lr, n_steps=g)
                run_train.launches += 1
                snap = None
                if verify_every and run_train.launches % verify_every == 0:
                    snap = (eng.params.detach().clone(), eng.m.clone(), eng.v.clone(), eng.global_step)
                replay()
                if snap is not None:
                    _verify_launch(eng, snap, replay.batches.clone(), g, lr)
                    run_train.verified_launches += 1
                logs.append(replay.tail_log.clone())
                last_x, last_rows = replay.batches[g - 1], replay.rows[g - 1]
            else:
                sx, replay = eng.capture_train_step(B, lr=lr, all_reduce=True, n_steps=g)
                xs = sx if g > 1 else sx.unsqueeze(0)
                for i in range(g):                          # this launch's batches, binarised on the device
                    last_rows = ds.next_rows(B)
                    binarize(ds.pixels, rows=last_rows, seed=bseed, step=eng.global_step + i, out=xs[i], out_row0=rank * B)
                replay()
                logs.append(replay.tail_log.clone())
                last_x = xs[g - 1]
            n -= g

    stop = False
    while eng.global_step <= config.max_steps and not stop:  # `<=`: the reference runs one extra step (runners.py:231)
        n = min(every - eng.global_step % every, config.max_steps + 1 - eng.global_step)
        run(n)
        tails = torch.cat(logs).cpu()                       # the summary's host sync
        logs = []
        vals = (tails[:, 0] / tails[:, 4]).tolist()
        base = eng.global_step - len(vals)
        fault = getattr(config, "fault_hook", None)         # (tests: called with the engine after every summary block)
        timed_out = bool(eng.handoff_timeouts())            # the explicit flag of a hand-off that gave up waiting
        nonfinite = not all(np.isfinite(vals))
        if world > 1:                                       # every rank takes the same branch (the tails are all-reduced,
            flag = torch.tensor([int(timed_out)], device=eng.device)     # the error word is per device)
            parallel.all_reduce_flat(flag, op=parallel.dist.ReduceOp.MAX)
            timed_out = bool(flag.item())
        if nonfinite and not timed_out:
            # no hand-off gave up: the loss itself left the finite range (divergence, bad input).  Changing the schedule
            # would re-run the same arithmetic; stop here with the last good checkpoint, like a failed sess.run.
            # (A deliberate deviation, INTEGRATION.md "Non-finite losses": the reference's MonitoredTrainingSession has no
            #  NanTensorHook and would keep stepping on NaN parameters.  No checkpoint is written here -- the parameters
            #  already went through the non-finite update -- the one on disk is from a finite step.)
            bad = [base + i + 1 for i, v in enumerate(vals) if not np.isfinite(v)]
            last_good = bad[0] - 1
            raise RuntimeError(f"non-finite training loss at step(s) {bad[:8]} with no hand-off timeout on any rank: "
                               f"training diverged (or the inputs are not finite); last step with a finite loss: {last_good}; "
                               f"the last good checkpoint ({_ckpt(config)}) was kept")
        good = [(base + i + 1, v) for i, v in enumerate(vals) if np.isfinite(v)]      # (true step index, loss)
        if timed_out:
            # A hand-off of the fused schedule timed out: something else holds part of the chip (a co-tenant, a
            # partitioned device).  The poisoned steps carry NaN losses and the optimizer SKIPPED them (params, m, v
            # untouched), so the model is intact.  Like MonitoredTrainingSession recovering from a failed step
            # (scripts/runners.py:222-232) the loop goes on -- once: re-captured on the schedule without mutual waits.
            if eng.safe_schedule:
                raise RuntimeError(f"training step poisoned between steps {base + 1} and {eng.global_step} although the "
                                   f"schedule without mutual waits is in use (hand-off timeouts: {eng.handoff_timeouts()}; "
                                   f"losses finite: {not nonfinite}); the last good checkpoint was kept")
            bad = [i for i, v in enumerate(vals) if not np.isfinite(v)]
            rewind = bool(bad) and bad == list(range(bad[0], len(vals)))      # the poisoned steps are the block's tail:
            if rewind:                                                        # run them again (new batches, same noise keys)
                eng.global_step = base + bad[0]
                eng.step_dev.fill_(eng.global_step)
            if rank == 0:
                print(f"[run_train] hand-off timeout: {len(bad)} step(s) of {base + 1}..{base + len(vals)} skipped by the "
                      f"optimizer; continuing from step {eng.global_step} on the schedule without mutual waits", flush=True)
            eng.use_safe_schedule()                         # (a field of THIS engine's dims: no process-wide state)
            run_train.degraded = True
        for step_i, v in good:                              # EarlyStoppingHook sees every applied step's (all-reduced) loss
            stop = hook.after_run(step_i, v) or stop        # under its TRUE step index (skipped steps leave gaps)
        vals = [v for _, v in good]
        if fault is not None:
            fault(eng)
        if not vals:
            continue
        if rank == 0 and (eng.global_step % every == 0 or eng.global_step > config.max_steps):
            rate = (eng.global_step - s0) / max(time.time() - t0, 1e-9)
            msg = f"Step {eng.global_step}, loss: {vals[-1]:f}  ({rate:.1f} global_step/sec)"
            if config.model == "gmvae" and ds.labels is not None:
                q = model.encoder_y(last_x).distribution.logits
                acc = utils.cluster_acc(q, ds.labels[last_rows.long()], config.mixture_components)
                msg += f"  cluster_acc {acc.item():.4f}"
            print(msg, flush=True)
        if stop and rank == 0:
            print("[Early Stopping Criterion Satisfied]")
        if rank == 0 and time.time() - last_save > 120:     # save_checkpoint_secs=120 (runners.py:226)
            _save_checkpoint(model, _ckpt(config))
            last_save = time.time()
    if rank == 0:
        _save_checkpoint(model, _ckpt(config))
    return model


@torch.no_grad()
def run_eval(config):
    """scripts/runners.py:235-459 without the plots.  Reports the true per-example loss and, under a
    different name, the reference's figure (sum of per-batch MEANS / number of examples,
    runners.py:298,330-335 -- i.e. roughly loss / batch_size).  Like the reference it WAITS for a checkpoint
    (scripts/utils.py:100-111, 60 s polls; config.checkpoint_poll_seconds / checkpoint_max_wait adjust that)."""
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(select_device(config, local))
    data_dim = int(getattr(config, "data_dim", 784))
    model = create_model(config, data_dim)
    wait_for_checkpoint(_ckpt(config), float(getattr(config, "checkpoint_poll_seconds", 60.0)),
                        getattr(config, "checkpoint_max_wait", None))
    model.load_state_dict(torch.load(_ckpt(config), map_location="cpu"))
    eng = model._engine
    tot = torch.zeros(5, device=eng.device)
    ref_sum, n_batches, codes, labs = 0.0, 0, [], []
    for images, labels in create_dataset(config, config.split, shuffle=False, repeat=False):
        o = eng.forward(images)
        tot += o["tail"][:5]
        ref_sum += (o["tail"][0] / o["tail"][4]).item()
        n_batches += 1
        # z = model.transform(flat_inputs) (runners.py:274): the VAE's MEAN code (vae.py:108-114), the GMVAE's SAMPLED
        # code (gmvae.py:140-149) -- the forward pass's z is exactly that sample
        codes.append(o["z"] if config.model == "gmvae" else model.transform(images))
        labs.append(labels)
    if world > 1:
        parallel.all_reduce_flat(tot)
    n = tot[4].item()
    res = {f"{config.split}/loss_per_example": tot[0].item() / n, f"{config.split}/nll": tot[1].item() / n,
           f"{config.split}/kl_div_z": tot[2].item() / n, f"{config.split}/nent": tot[3].item() / n,
           f"{config.split}/reference_misnormalised_loss_per_example": ref_sum / n, "examples": int(n)}
    if rank == 0:
        for k, v in res.items():
            print(f"{k}: {v}")
    # The tensors the reference's evaluation graph also produces (scripts/runners.py:274-292) and hands to its plots
    # (which are out of scope): the latent state and labels over the split, `num_samples` draws from the prior and
    # `num_generations` decoded prior draws; for the GMVAE additionally the decoded draws of ONE random component k,
    # stacked on the unconditional ones (runners.py:285-292).
    img_shape = (28, 28, 1) if data_dim == 784 else (data_dim, 1, 1)
    res["latent_state"] = torch.cat(codes) if codes else None
    res["labels"] = torch.cat(labs) if labs else None
    res["samples"] = model.generate_samples(num_samples=int(config.num_samples))
    sample_images = utils.unflatten_tensor(model.generate_sample_images(num_samples=int(config.num_generations)), img_shape)
    if config.model == "gmvae":
        k = int(np.random.randint(0, high=config.mixture_components))
        samples_k = model.generate_samples(num_samples=int(config.num_generations) * int(config.mixture_components), clusters=[k])
        sample_images_k = utils.unflatten_tensor(model.generate_sample_images(samples_k, name="sample_images_k"), img_shape)
        sample_images = torch.stack((sample_images, sample_images_k), dim=0)
        res["sampled_cluster"] = k
    res["sample_images"] = sample_images
    return res
