"""`fit()`: the counterpart of `Trainer.fit(model=model_setup, datamodule=dm)` + `to_torchscript` of the reference's
train_model.py:27-61 for this engine -- no Lightning needed.  It drives exactly the pieces the reference's Trainer
drives on a ModelMeta: configure_optimizers (SAMSGD(Adam) + MultiStepLR([40], 0.1)), one optimisation step per
batch, `training_epoch_end`, `validation_step` / `validation_epoch_end` under no_grad, `scheduler.step()` once per
epoch, and at the end `to_torchscript(path)`.

Batches are host tensors as a DataLoader yields them: (x, y, gt_bbxs) with x either uint8 frames (B,3,H,W) -- fed
through `U8BatchFeeder` (pinned staging or the loader's own pinned tensors, copy stream, /255 or bilinear resize on the
device, three slots: the copy of batch i+1 runs under step i) -- or float32 in [0,1]; y the encoded targets (B,5,S,S).  The optimisation step is `fused_train_step` (forward +
YoloLoss + backward + all-reduce + Adam as direct kernel launches); its step outputs (loss, total_iou, total_recall,
total_precision) have the meaning of `ModelMeta.training_step`'s (models/ModelMeta.py:115-227).
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional

import torch

from .datasets.feed import U8BatchFeeder
from .hostinfo import limit_host_threads


def _epoch(model_meta, batches, train: bool, feeder_cache: dict, on_step: Optional[Callable]):
    model = model_meta.model
    dev = next(model.parameters()).device
    size = tuple(model.input_shape[1:])
    outs: List[dict] = []
    it = iter(batches)
    pending = None                    # (x_dev, y_dev, token) of the batch whose copy is in flight

    def stage(batch):
        x, y = batch[0], batch[1]
        if x.dtype == torch.uint8 and not x.is_cuda:
            key = (tuple(x.shape), tuple(y.shape))
            fd = feeder_cache.get(key)
            if fd is None:
                fd = feeder_cache[key] = U8BatchFeeder(tuple(x.shape), size, dev, target_shape=tuple(y.shape))
            fd.submit(x, y.float())
            return ("feeder", fd)
        return ("direct", (x.to(dev, non_blocking=True).float(), y.to(dev, non_blocking=True).float()))

    nxt = next(it, None)
    if nxt is not None:
        pending = stage(nxt)
    step = 0
    while pending is not None:
        kind, payload = pending
        if kind == "feeder":
            x_d, y_d, tok = payload.get()
        else:
            (x_d, y_d), tok = payload, None
        # the step of THIS batch is enqueued first; only then is the next batch staged.  Staging may block the host on the
        # slot's previous consumer (feed.py: `free.synchronize()`), and with the step already queued the GPU keeps working
        # through that wait; with 3 slots the reused slot belongs to step i-2, which has normally finished.
        if train:
            model.train()
            lsum, y_hat, tot = model_meta.fused_train_step(x_d, y_d, with_metrics=True)
            out = {"loss": lsum.detach().reshape(()), "total_iou": tot[0], "total_recall": tot[1], "total_precision": tot[2]}
        else:
            model.eval()
            with torch.no_grad():
                out = model_meta.validation_step((x_d, y_d, None), step)
        if tok is not None:
            U8BatchFeeder.release(tok)
        nxt = next(it, None)
        following = stage(nxt) if nxt is not None else None          # the next batch's copy overlaps this step
        outs.append(out)
        if on_step is not None:
            on_step(step, train, out)
        step += 1
        pending = following
    return outs


def fit(model_meta, train_batches: Iterable, val_batches: Optional[Iterable] = None, epochs: int = 1,
        torchscript_path: Optional[str] = None, on_step: Optional[Callable] = None) -> dict:
    """Train `model_meta` (a ModelMeta) for `epochs` passes over `train_batches` (re-iterable), validating on
    `val_batches` after every epoch.  Returns {"train": [per-epoch metrics], "val": [...], "scripted": module or None}."""
    limit_host_threads()                  # 256 OpenMP workers on a 16-core share stall the thread that feeds the GPU
    optimizers, schedulers = model_meta.configure_optimizers()
    sched = schedulers[0]
    feeder_cache: dict = {}
    hist = {"train": [], "val": [], "scripted": None}
    for epoch in range(epochs):
        model_meta.current_epoch = epoch
        if val_batches is not None:
            # Lightning runs the validation loop inside the training epoch and calls validation_epoch_end first, which is
            # why the reference's training line in the log quotes the validation metrics of the same epoch (ModelMeta.py:268-289)
            pass
        tr = _epoch(model_meta, train_batches, True, feeder_cache, on_step)
        if val_batches is not None:
            va = _epoch(model_meta, val_batches, False, feeder_cache, on_step)
            hist["val"].append(model_meta.format_metrics(va, training=False))
        hist["train"].append(model_meta.format_metrics(tr, training=True))
        sched.step()
    model_meta.model.train()
    if torchscript_path is not None:
        hist["scripted"] = model_meta.to_torchscript(torchscript_path)
    return hist
