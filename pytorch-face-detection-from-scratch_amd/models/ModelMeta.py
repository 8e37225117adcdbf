"""`ModelMeta` with the reference's surface (models/ModelMeta.py:85-322): forward,
configure_optimizers, step / training_step / validation_step / test_step, format_metrics and
the *_epoch_end hooks.  Subclasses LightningModule when pytorch_lightning is importable, else
nn.Module with the few attributes the hooks use.

Two ways to train:
  * Lightning-style: `training_step` returns {"loss": tensor-with-grad_fn, ...}; the caller
    runs loss.backward() and optimizer.step() (autograd bridge ConvStackFn).
  * `fused_train_step(x, y)`: forward + loss + backward + (all-reduce) + Adam as direct kernel
    launches, no autograd bookkeeping -- what bench.py times.
"""
from pathlib import Path

import torch

try:                                                     # pragma: no cover - not in this image
    from pytorch_lightning import LightningModule as _Base
    _HAVE_PL = True
except Exception:                                        # noqa: BLE001
    _Base = torch.nn.Module
    _HAVE_PL = False

from .. import hotpath as hp
from ..dataparallel import GradBucketReducer, sync_parameters
from ..losses.YoloLoss import yolo_loss_batch
from ..optim import SAMSGD


class ModelMeta(_Base):
    def __init__(self, model, lr=1e-4, pretrained=False, log_path=Path("out.log"), *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.model = model
        self.lr = lr
        self.automatic_optimization = True
        self.log_path = log_path
        self.opt = None
        self.epoch_metrics = None
        if not _HAVE_PL:
            self.current_epoch = 0
        self._logged = {}
        self._reducer = None

    if not _HAVE_PL:
        def log(self, name, value, **kwargs):            # Lightning's self.log stand-in
            self._logged[name] = value

    def forward(self, x):
        return self.model(x)

    def configure_optimizers(self):
        optimizer = SAMSGD(self.parameters(), lr=self.lr)
        optimizer.on_params_updated = self.model.engine.mark_params_dirty      # (MobilenetV3Backbone: its inference pack)
        self.opt = optimizer
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[40], gamma=0.1)
        return [optimizer], [scheduler]

    # ------------------------------------------------------------------ metrics (ModelMeta.py:170-218)
    @torch.no_grad()
    def _metrics(self, y_hat, y):
        red = self.model.reduce_bounding_boxes
        gt, gc = red.forward_batch(y)
        pr, pc = red.forward_batch(y_hat.detach())
        _, tot = hp.step_metrics(gt, gc, pr, pc)
        return tot                                        # [sum IoU, recall, precision] / B

    def step(self, batch, batch_idx, validation=False):
        x, y, gt_bbxs = batch
        y_hat = self.forward(x)
        loss = yolo_loss_batch(y_hat, y)                  # batch SUM (:173-176, :215 commented out)
        tot = self._metrics(y_hat, y)
        step_outputs = {"loss": loss, "total_iou": tot[0], "total_recall": tot[1], "total_precision": tot[2]}
        self.log("step_loss", loss, prog_bar=True, logger=True, on_step=True)
        return step_outputs

    def training_step(self, batch, batch_idx):
        return self.step(batch, batch_idx)

    def validation_step(self, batch, batch_idx):
        return self.step(batch, batch_idx, validation=True)

    def test_step(self, batch, batch_idx):
        return self.step(batch, batch_idx, validation=True)

    # ------------------------------------------------------------------ fused step (no autograd)
    def fused_train_step(self, x, y, with_metrics: bool = False):
        """One optimisation step on (x (N,3,H,W) f32 in [0,1], y (N,5,S,S)).  Under
        torch.distributed each rank passes ITS shard; gradients are SUM-all-reduced.
        Returns (loss_sum (1,), y_hat, metrics or None) -- GPU tensors, no host sync."""
        if self.opt is None:
            self.configure_optimizers()
        if hasattr(self.model, "_train_engine"):
            return self._fused_train_step_mobilenet(x, y, with_metrics)
        model, eng = self.model, self.model.engine
        names, params = model.named_stack_params()
        opt = self.opt
        sp = opt._space()
        if [id(p) for p in sp.params] != [id(p) for p in params]:
            raise RuntimeError("optimizer parameter order differs from the conv stack's")
        P = {n: p.data for n, p in zip(names, params)}
        G = {n: sp.view(sp.grad, i) for i, n in enumerate(names)}
        if self._reducer is None or self._reducer.flat.data_ptr() != sp.grad.data_ptr():
            split_block = min(2, len(model.residual_blocks))
            split = sp.offsets[names.index(f"residual_blocks.{split_block}.conv1.weight")] \
                if split_block < len(model.residual_blocks) else sp.offsets[names.index("out.weight")]
            self._reducer = GradBucketReducer(sp.grad, split)
            self._split_block = split_block
            if self._reducer.enabled:
                # start-of-training hand-shake: every rank continues from rank 0's weights (checksummed)
                sync_parameters(sp.flat)
                eng.mark_params_dirty()
                P = {n: p.data for n, p in zip(names, params)}
        red = self._reducer
        masks = model._draw_masks(x.shape[0], x.device) if model.training else None
        # PoolResnet geometry: head forward + yolo_loss + their gradients run as ONE fused kernel at the end of forward
        y_hat, saved = eng.forward(x, P, masks, save=True, loss_targets=y if eng.head_loss_fusable() else None, G=G)
        if saved.get("loss") is not None:
            (_, lsum), dy = saved["loss"], None
        else:
            _, lsum, dy = hp.yolo_loss_fwd_bwd(y_hat, y, want_grad=True)
        eng.backward(saved, dy, P, G,
                     after_block=(lambda k: red.launch_tail() if k == self._split_block else None) if red.enabled else None)
        if red.enabled:
            red.launch_head()
            red.wait()
        opt.step(grads_in_flat=True)
        metrics = self._metrics(y_hat, y) if with_metrics else None
        return lsum, y_hat, metrics

    def _fused_train_step_mobilenet(self, x, y, with_metrics: bool = False):
        """The same step for `MobilenetV3Backbone` (mobilenet_train.py): training-mode forward (BatchNorm batch statistics,
        running statistics updated), yolo_loss, the hand-written backward writing straight into the optimiser's flat gradient
        buffer, SUM all-reduce under torch.distributed, Adam -- no autograd graph, no per-parameter gradient copies (the
        Lightning-style training_step + loss.backward() + optimizer.step() path stays available and gives the same update)."""
        model, opt = self.model, self.opt
        if not x.is_cuda:
            raise hp.N.FdetError("the MobileNet training path runs on the GPU only (no CPU fallback): move model and input to cuda")
        params = dict(model.named_parameters())
        names = list(params.keys())
        sp = opt._space()
        if [id(p) for p in sp.params] != [id(params[n]) for n in names]:
            raise RuntimeError("optimizer parameter order differs from the model's")
        P = {n: p.data for n, p in params.items()}
        P.update({n: b for n, b in model.named_buffers()})
        G = {n: sp.view(sp.grad, i) for i, n in enumerate(names)}
        if self._reducer is None or self._reducer.flat.data_ptr() != sp.grad.data_ptr():
            self._reducer = GradBucketReducer(sp.grad, sp.offsets[len(names) // 2])
            if self._reducer.enabled:
                sync_parameters(sp.flat)
                P.update({n: p.data for n, p in params.items()})
        red = self._reducer
        was_training = model.training
        if not was_training:
            raise RuntimeError("fused_train_step: put the model in train() mode (BatchNorm batch statistics)")
        y_hat, saved = model._train_engine.forward_train(x, P)
        _, lsum, dy = hp.yolo_loss_fwd_bwd(y_hat, y, want_grad=True)
        model._train_engine.backward(saved, dy, P, G)
        if red.enabled:
            red.launch_tail()
            red.launch_head()
            red.wait()
        opt.step(grads_in_flat=True)
        metrics = self._metrics(y_hat, y) if with_metrics else None
        return lsum, y_hat, metrics

    # ------------------------------------------------------------------ export (train_model.py:61)
    def to_torchscript(self, file_path=None, method="script", example_inputs=None, **kwargs):
        """`LightningModule.to_torchscript` (train_model.py:61) -- overridden with Lightning installed too: its default
        would torch.jit.script the whole module, whose forward is ctypes / HIP launches.  Returns the scripted inference
        path of the wrapped model (custom `fdet::` operators, torchscript.py), saved to `file_path` when given."""
        if method != "script":
            raise ValueError("only method='script' is supported (tracing records nothing of a custom-operator forward)")
        from ..torchscript import to_torchscript
        return to_torchscript(self.model, file_path)

    # ------------------------------------------------------------------ epoch hooks (ModelMeta.py:241-322)
    def format_metrics(self, epoch_outputs, training=True):
        step_str = "training" if training else "validation"
        m = {}
        m["loss"] = torch.mean(torch.tensor([float(e["loss"]) for e in epoch_outputs]))
        m["total_iou"] = torch.mean(torch.tensor([float(e["total_iou"]) for e in epoch_outputs]))
        m["total_recall"] = torch.mean(torch.tensor([float(e["total_recall"]) for e in epoch_outputs]))
        m["total_precision"] = torch.mean(torch.tensor([float(e["total_precision"]) for e in epoch_outputs]))
        m["f1_score"] = 2 * m["total_precision"] * m["total_recall"] / (m["total_precision"] + m["total_recall"])  # NaN at 0/0 (Q16)
        self.log("loss", m["loss"], prog_bar=True, logger=True, on_epoch=True)
        for key, nm in (("loss", "loss"), ("total_iou", "iou"), ("total_recall", "recall"),
                        ("total_precision", "precision"), ("f1_score", "f1_score")):
            self.log(f"{step_str} {nm}", m[key], prog_bar=True, logger=True, on_epoch=True)
        if training:
            print(f"\nEpoch: {self.current_epoch}, lr: {self.opt.param_groups[0]['lr']}", end=" ")
        print(f"\n{step_str}, loss: {m['loss']:5.3f}", end=" ")
        if not training:
            self.epoch_metrics = m
        else:
            em = self.epoch_metrics or {k: float("nan") for k in m}
            with Path(self.log_path).open("a") as fp:
                fp.write(f"\nEpoch: {self.current_epoch}, lr: {self.opt.param_groups[0]['lr']} ")
                fp.write(f"training, loss: {m['loss']:5.3f}, iou: {m['total_iou']:5.3f},"
                         f"recall {m['total_recall']:5.3f}, precision {m['total_precision']:5.3f}"
                         f", f1_score {m['f1_score']:5.3f} ")
                fp.write(f"validation, loss: {em['loss']:5.3f}, iou: {em['total_iou']:5.3f},"
                         f" recall {em['total_recall']:5.3f}, precision {em['total_precision']:5.3f}"
                         f", f1_score {em['f1_score']:5.3f} ")
        return m

    def training_epoch_end(self, training_epoch_outputs):
        self.format_metrics(training_epoch_outputs, training=True)

    def validation_epoch_end(self, validation_epoch_outputs):
        self.format_metrics(validation_epoch_outputs, training=False)

    def test_epoch_end(self, validation_epoch_outputs):
        self.format_metrics(validation_epoch_outputs, training=False)
