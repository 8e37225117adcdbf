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


def rank_world(group=None):
    """(rank, world) of this process; (0, 1) outside torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def dropout_stream(calls: int, n_local: int, group=None):
    """(base, first_image) of the dropout counter for the `calls`-th training forward of a rank that holds `n_local`
    images (equal shards: the global batch is world * n_local; rank r owns images [r*n_local, (r+1)*n_local)).
    `base` does not depend on the batch size (ranges of different calls must never overlap):
    call k owns counters [k << 40, (k+1) << 40)."""
    rank, _ = rank_world(group)
    return int(calls) << 40, rank * int(n_local)


def sync_parameters(flat_param: torch.Tensor, group=None, check: bool = True) -> None:
    """Start-of-training hand-shake: broadcast rank 0's flat parameter buffer (ranks then hold bit-identical
    weights whatever their seeds were) and, with `check`, verify it by an all-reduced checksum."""
    rank, world = rank_world(group)
    if world == 1 and not dp_active(group):
        return
    dist.broadcast(flat_param, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if check:
        s = flat_param.double().sum().reshape(1)
        lo, hi = s.clone(), s.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
        if float(lo) != float(hi):
            raise RuntimeError(f"rank {rank}: parameters differ across ranks after the broadcast "
                               f"(checksums {float(lo)!r} .. {float(hi)!r})")


def shard_range(batch: int, rank: int, world: int):
    """Images [lo, hi) of a global batch owned by `rank` (contiguous, remainder to low ranks)."""
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def dp_forced() -> bool:
    """FDET_DP_FORCE=1: run the gradient exchange even in a one-rank process group (the all-reduce is then the identity).
    This is how the RCCL path -- init with device_id, async all-reduce on slices of the flat gradient, stream-ordered
    wait -- is exercised on a one-GPU box (tests/test_gpu_rccl.py)."""
    import os
    return os.environ.get("FDET_DP_FORCE", "0") == "1"


def dp_active(group=None) -> bool:
    """A process group exists and has more than one rank (or the exchange is forced, see dp_forced)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or dp_forced()


class GradBucketReducer:
    def __init__(self, flat_grad: torch.Tensor, split_offset: int, group=None):
        """`split_offset`: index into flat_grad where the late (low-resolution) bucket starts."""
        if not (0 <= split_offset <= flat_grad.numel()):
            raise ValueError("split_offset outside the gradient buffer")
        self.flat = flat_grad
        self.split = split_offset
        self.group = group
        self._pending: List = []
        self.enabled = dp_active(group)
        self.timing = None           # optional [(event before wait, event after wait)] list: exposed all-reduce time

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
        """Make the current stream wait for the launched all-reduces (no host block).  With `timing` set to a list, a
        pair of HIP events brackets the wait on the current stream: what the step could NOT hide behind backward."""
        ev = None
        if self.timing is not None and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._pending:
            w.wait()
        self._pending = []
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)


def allreduce_scalars(t: torch.Tensor, group=None) -> torch.Tensor:
    """SUM of per-rank step scalars (loss, metric sums)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
