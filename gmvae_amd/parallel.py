"""Data-parallel plumbing: one process per GPU, ONE all-reduce per step.

The reference is single-device (scripts/runners.py:193 pins one GPU; SURVEY.md
2.2).  The loss is a batch mean (scripts/gmvae.py:254,258,262), so gradients are
sums over samples: rank r takes rows [r*B/G, (r+1)*B/G) of the global batch,
computes gradient SUMS into the flat [P + TAIL] buffer (gmvae_step), a single
all-reduce(SUM) -- RCCL over xGMI via torch.distributed backend "nccl" -- makes
the buffer global (loss sums and the sample count ride in the tail), and every
rank applies the identical TF-Adam update scaled by 1/count, so replicas stay
bit-identical without ever broadcasting parameters again.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist

TAIL = 8


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:                        # GMVAE_DIST_BACKEND=gloo: several ranks on ONE device (tests; RCCL refuses that)
            backend = os.environ.get("GMVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row range of `rank`; the first n_rows % world ranks take one extra row
    (the reference's last partial batch, scripts/runners.py:51, need not divide evenly)."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_reduce_flat(buf: torch.Tensor, op=None) -> torch.Tensor:
    """The step's single collective: SUM (or `op`) of the flat [P + TAIL] buffer over all ranks, in place.  RCCL reduces
    device tensors where they lie; over gloo (CPU collectives: tests with several ranks on one device) a device tensor is
    staged through the host."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        op = dist.ReduceOp.SUM if op is None else op
        if buf.is_cuda and dist.get_backend() != "nccl":
            host = buf.detach().cpu()
            dist.all_reduce(host, op=op)
            buf.copy_(host.to(buf.device))
        else:
            dist.all_reduce(buf, op=op)
    return buf


def grad_scale(buf: torch.Tensor, P: int) -> torch.Tensor:
    """1 / global sample count, read from the all-reduced tail (a tensor: no host sync)."""
    return 1.0 / buf[P + 4]


def broadcast_state(tensors, ints=(), src: int = 0):
    """Start-of-training replica sync (and after a checkpoint restore on rank 0 only): every rank takes rank `src`'s
    tensors (parameters, Adam moments ...) IN PLACE and returns rank `src`'s integers (global step, noise seed ...).
    The reference's default `--random_seed=None` (scripts/run_gmvae.py:30) seeds each process from entropy, so without
    this the ranks would apply identical all-reduced gradients to different parameters.  Works with device tensors
    over RCCL (backend "nccl") and with CPU or device tensors over gloo (staged through the host)."""
    ints = [int(i) for i in ints]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return ints
    on_dev = dist.get_backend() == "nccl"
    meta = torch.tensor(ints if ints else [0], dtype=torch.int64)
    if on_dev:
        meta = meta.cuda()
    dist.broadcast(meta, src=src)
    with torch.no_grad():
        for t in tensors:
            if t.is_cuda and not on_dev:
                buf = t.detach().cpu()
                dist.broadcast(buf, src=src)
                t.copy_(buf.to(t.device))
            else:
                dist.broadcast(t.detach(), src=src)
    return [int(v) for v in meta.tolist()][:len(ints)]


def assert_replicas_identical(params: torch.Tensor) -> bool:
    """Debug check: max |params - params_rank0| == 0 on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return True
    ref = params.detach().clone()
    dist.broadcast(ref, src=0)
    return bool((ref == params.detach()).all().item())
