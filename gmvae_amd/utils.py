"""Host helpers mirroring scripts/utils.py where the hot path touches them."""
from __future__ import annotations

import torch

from . import _lib as L


def flatten_tensor(inputs, shape=None, name="flattened"):
    """scripts/utils.py:140-141: [B,28,28,1] -> [B,784]."""
    return inputs.reshape(inputs.shape[0], -1)


def unflatten_tensor(inputs, shape, name="unflattened"):
    """scripts/utils.py:144-145."""
    return inputs.reshape(-1, shape[0], shape[1], shape[2])


def entropy(logits, targets):
    """scripts/utils.py:165-170: -sum(targets * log_softmax(logits)) per row."""
    return -(targets * torch.log_softmax(logits, -1)).sum(1)


def cluster_acc_from_hist(hist: torch.Tensor) -> torch.Tensor:
    """Clustering accuracy from the [K, n_labels] cluster x label histogram: matches = sum_k max_l hist[k, l]
    (every sample of a cluster that carries the cluster's majority label).  Histograms add across data-parallel
    ranks, so this is the multi-GPU definition of scripts/utils.py:173-191."""
    hist = hist.to(torch.int64)
    n = hist.sum()
    return hist.max(dim=1).values.sum().to(torch.float32) / n.clamp(min=1).to(torch.float32)


def cluster_acc(logits, labels, no_components, n_labels: int = 10, all_reduce: bool = False):
    """scripts/utils.py:173-191 on the device (gmvae_cluster_acc).  Returns a 0-d tensor.  With all_reduce (and an
    initialised process group) the ranks' histograms are summed first: the accuracy of the global batch."""
    dev = L.require_gpu()
    logits = logits.to(dev, torch.float32).contiguous()
    labels = torch.as_tensor(labels).to(dev, torch.int64).contiguous()
    B, K = logits.shape
    if K != no_components:
        raise ValueError("logits second dim must equal no_components")
    if not all_reduce:           # (ranks must agree on the histogram's shape: no data-dependent widening there)
        n_labels = max(n_labels, int(labels.max().item()) + 1) if labels.numel() else n_labels
    scratch = torch.empty(K * n_labels + B, dtype=torch.int32, device=dev)
    acc = torch.empty(1, dtype=torch.float32, device=dev)
    L.check(L.lib.gmvae_cluster_acc(L.ptr(logits), L.ptr(labels), B, K, n_labels, L.ptr(scratch), L.ptr(acc),
                                    L.current_stream()), "gmvae_cluster_acc")
    if all_reduce:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            hist = scratch[:K * n_labels].clone().view(K, n_labels)     # the kernel's histogram: ranks' counts add
            from . import parallel
            parallel.all_reduce_flat(hist)
            return cluster_acc_from_hist(hist)
    return acc[0]


class EarlyStoppingHook:
    """scripts/utils.py:13-57: stop when the training-batch loss has not improved by a
    relative `threshold` for `max_steps` consecutive steps; counters reset if the
    global step goes backwards (recovery)."""

    def __init__(self, max_steps=100, threshold=0.001):
        self._max_steps, self._threshold = max_steps, threshold
        self._last_step = -1
        self._steps = 0
        self._prev_loss = None

    def after_run(self, curr_step: int, curr_loss: float) -> bool:
        """Returns True when training should stop."""
        self._steps += 1
        if self._last_step == -1 or self._last_step > curr_step:
            self._last_step = curr_step
            self._steps = 0
            self._prev_loss = None
            return False
        self._last_step = curr_step
        if self._prev_loss is None or curr_loss < (self._prev_loss - self._prev_loss * self._threshold):
            self._prev_loss = curr_loss
            self._steps = 0
        return self._steps >= self._max_steps
