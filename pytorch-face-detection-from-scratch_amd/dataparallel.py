"""Data-parallel gradient exchange: one process per GPU, `torch.distributed` (backend "nccl"
is RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-process; this layer is new.  Every image is
independent in forward, loss and metrics (no BatchNorm), so the batch shards across ranks and
the only exchange per step is a SUM all-reduce of the flat fp32 gradient buffer -- SUM, not
mean, because the reference's loss is the batch SUM (models/ModelMeta.py:176): N ranks on
B/N images each reproduce the single-process gradients of the concatenated batch.

The gradient buffer is laid out in parameter order (stem | block0 | ... | block9 | head).
Backward produces gradients in reverse, so the buffer is cut into two contiguous buckets:
the tail (blocks >= split, head) is reduced as soon as block `split` has been back-propagated
and overlaps with the expensive high-resolution blocks and the stem; the head bucket follows.
At 3 MB total the exchange is latency-bound, so two buckets are enough.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


def shard_range(batch: int, rank: int, world: int):
    """Images [lo, hi) of a global batch owned by `rank` (contiguous, remainder to low ranks)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucketReducer:
    def __init__(self, flat_grad: torch.Tensor, split_offset: int, group=None):
        """`split_offset`: index into flat_grad where the late (low-resolution) bucket starts."""
        if not (0 <= split_offset <= flat_grad.numel()):
            raise ValueError("split_offset outside the gradient buffer")
        self.flat = flat_grad
        self.split = split_offset
        self.group = group
        self._pending: List = []
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def launch_tail(self):
        """Gradients of [split, end) are final: start their all-reduce (async)."""
        if self.enabled and self.split < self.flat.numel():
            self._pending.append(dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def launch_head(self):
        if self.enabled and self.split > 0:
            self._pending.append(dist.all_reduce(self.flat[:self.split], op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []


def allreduce_scalars(t: torch.Tensor, group=None) -> torch.Tensor:
    """SUM of per-rank step scalars (loss, metric sums)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
