"""The reference's input pipeline on the device (scripts/runners.py:21-62 `create_dataset`).

The reference maps every MNIST example through ``image = pixel / 255.; image = image < uniform`` (dynamic
binarisation, re-drawn on every pass; note that it sets a pixel with probability 1 - intensity), batches, and
shuffles.  Here the raw uint8 pixels stay resident in HBM (60000 x 784 = 47 MB), an epoch permutation lives on the
device, and `gmvae_binarize` (HIP) produces each step's uint8 [B, D] batch in the layout `gmvae_step` takes:
no host -> device copy per step.  The TFDS download itself is host plumbing and out of scope: pixels come in as a
tensor.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def binarize(pixels: torch.Tensor, rows: Optional[torch.Tensor] = None, row0: int = 0, batch: Optional[int] = None,
             seed: int = 0, step: int = 0, out: Optional[torch.Tensor] = None, out_row0: int = 0) -> torch.Tensor:
    """x[b, :] = (pixels[rows[b], :] / 255 < U) as uint8 0/1 (runners.py:44-47).  `rows`: int32 device tensor of
    source rows, or None for rows row0 .. row0+batch-1.  The uniforms are Philox4x32-10 keyed by (seed, step) and the
    element's position in the global batch (out_row0 = this shard's first row: rank * B under data parallelism)."""
    dev = L.require_gpu()
    if pixels.dtype != torch.uint8 or pixels.dim() != 2 or not pixels.is_cuda or not pixels.is_contiguous():
        raise ValueError("pixels must be a contiguous uint8 [N, D] tensor on the GPU")
    N, D = pixels.shape
    if rows is not None:
        rows = rows.to(dev, torch.int32).contiguous()
        B = rows.numel()
    else:
        B = int(batch if batch is not None else N - row0)
    if out is None:
        out = torch.empty(B, D, dtype=torch.uint8, device=dev)
    L.check(L.lib.gmvae_binarize(L.ptr(pixels), N, L.ptr(rows) if rows is not None else None, int(row0), B, D,
                                 int(seed), int(step), None, L.ptr(out), int(out_row0), L.current_stream()),
            "gmvae_binarize")
    return out


class DeviceDataset:
    """uint8 pixels [N, ...] (+ labels) resident on the GPU; `next_batch(B)` returns a freshly binarised uint8
    [B, D] batch and its labels.  shuffle=True draws a new permutation per epoch on the device; every batch gets
    new uniforms (step counter), as the reference's `repeat()` after `map()` does."""

    def __init__(self, pixels, labels=None, shuffle: bool = True, seed: int = 0):
        dev = L.require_gpu()
        pixels = torch.as_tensor(pixels)
        self.pixels = pixels.reshape(pixels.shape[0], -1).to(dev, torch.uint8).contiguous()
        self.labels = None if labels is None else torch.as_tensor(labels).to(dev, torch.int64)
        self.N, self.D = self.pixels.shape
        self.shuffle, self.seed = shuffle, int(seed)
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(self.seed)
        self._perm = None
        self._pos = 0
        self._step = 0

    def _new_epoch(self):
        dev = self.pixels.device
        self._perm = (torch.randperm(self.N, device=dev, generator=self._gen) if self.shuffle
                      else torch.arange(self.N, device=dev)).to(torch.int32)
        self._pos = 0

    def next_rows(self, B: int) -> torch.Tensor:
        """Source rows of the next batch (int32, on the device); a short last batch wraps into the next epoch."""
        if self._perm is None:
            self._new_epoch()
        parts, need = [], B
        while need > 0:
            if self._pos >= self.N:
                self._new_epoch()
            take = min(need, self.N - self._pos)
            parts.append(self._perm[self._pos:self._pos + take])
            self._pos += take
            need -= take
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def next_batch(self, B: int, out: Optional[torch.Tensor] = None):
        rows = self.next_rows(B)
        x = binarize(self.pixels, rows=rows, seed=self.seed, step=self._step, out=out)
        self._step += 1
        return x, (None if self.labels is None else self.labels[rows.long()])
