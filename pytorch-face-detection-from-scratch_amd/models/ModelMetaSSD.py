"""`ModelMetaSSD` with the reference's surface (models/ModelMetaSSD.py:85-235): forward,
configure_optimizers, step / training_step / validation_step, plus `fused_train_step` (forward +
ssd_loss + backward + Adam as direct kernel launches).  Metrics as in step(): per-image
ReduceSSDBoundingBoxes on targets and predictions, box_iou hits at 0.5 (fdet_step_metrics)."""
from pathlib import Path

import torch

from .. import hotpath as hp
import torch.distributed as dist

from ..dataparallel import GradBucketReducer, dp_active, rank_world, sync_parameters
from ..losses.SSDLoss import ssd_loss
from ..optim import SAMSGD
from .ModelMeta import ModelMeta as _YoloMeta

try:                                                     # pragma: no cover - not in this image
    from pytorch_lightning import LightningModule as _Base
    _HAVE_PL = True
except Exception:                                        # noqa: BLE001
    _Base = torch.nn.Module
    _HAVE_PL = False


class ModelMetaSSD(_Base):
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
        self._reducer = None
        self._logged = {}

    if not _HAVE_PL:
        def log(self, name, value, **kwargs):            # Lightning's self.log stand-in
            self._logged[name] = value

    def forward(self, x):
        return self.model(x)

    def configure_optimizers(self):
        optimizer = SAMSGD(self.parameters(), lr=self.lr)
        optimizer.on_params_updated = self.model.engine.mark_params_dirty
        self.opt = optimizer
        scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[40], gamma=0.1)
        return [optimizer], [scheduler]

    @torch.no_grad()
    def _metrics(self, y_hat, y):
        red = self.model.reduce_bounding_boxes
        gt, gc = red.forward_batch(y)
        pr, pc = red.forward_batch(y_hat.detach())
        _, tot = hp.step_metrics(gt, gc, pr, pc)
        return tot

    def step(self, batch, batch_idx, validation=False):
        x, y, gt_bbxs = batch
        y_hat = self.forward(x)
        loss = ssd_loss(y_hat[:, :, 0], y_hat[:, :, 1:], y[:, :, 0], y[:, :, 1:], 10)      # :178
        tot = self._metrics(y_hat, y)
        out = {"loss": loss, "total_iou": tot[0], "total_recall": tot[1], "total_precision": tot[2]}
        self.log("step_loss", loss, prog_bar=True, logger=True, on_step=True)
        return out

    def training_step(self, batch, batch_idx):
        return self.step(batch, batch_idx)

    def validation_step(self, batch, batch_idx):
        return self.step(batch, batch_idx, validation=True)

    def test_step(self, batch, batch_idx):
        return self.step(batch, batch_idx, validation=True)

    # ------------------------------------------------------------------ epoch hooks (ModelMetaSSD.py:245-327)
    # line for line the hooks of models/ModelMeta.py:241-322 in the reference too: one implementation serves both
    def to_torchscript(self, file_path=None, method="script", example_inputs=None, **kwargs):
        """`LightningModule.to_torchscript` (train_model_ssd.py): the scripted inference module of the SSD mirror."""
        from ..torchscript import to_torchscript
        return to_torchscript(self.model, file_path)

    format_metrics = _YoloMeta.format_metrics
    training_epoch_end = _YoloMeta.training_epoch_end
    validation_epoch_end = _YoloMeta.validation_epoch_end
    test_epoch_end = _YoloMeta.test_epoch_end

    def fused_train_step(self, x, y):
        """One optimisation step on (x (N,3,480,480) in [0,1], y (N,4774,5)); returns (loss (1,), y_hat)."""
        if self.opt is None:
            self.configure_optimizers()
        model, eng = self.model, self.model.engine
        names, params = model.named_stack_params()
        sp = self.opt._space()
        if [id(p) for p in sp.params] != [id(p) for p in params]:
            raise RuntimeError("optimizer parameter order differs from the SSD stack's")
        P = {n: p.data for n, p in zip(names, params)}
        G = {n: sp.view(sp.grad, i) for i, n in enumerate(names)}
        import os
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("WORLD_SIZE > 1 but torch.distributed is not initialised: data-parallel SSD training "
                               "needs a process group (bench.py / torch.distributed.run set one up)")
        dp = dp_active()
        if dp and (self._reducer is None or self._reducer.flat.data_ptr() != sp.grad.data_ptr()):
            # one process per GPU: rank 0's parameters everywhere, then one SUM all-reduce of the flat gradient per step.
            # Re-created whenever the flat buffers were re-homed (model.to(), load_state_dict(assign=True), a resume):
            # a reducer bound to the OLD gradient buffer would reduce stale memory and let the ranks drift apart.
            sync_parameters(sp.flat)
            eng.mark_params_dirty()
            P = {n: p.data for n, p in zip(names, params)}
            self._reducer = GradBucketReducer(sp.grad, 0)
        masks = model._draw_masks(x.shape[0], x.device) if model.training else None
        y_hat, saved = eng.forward(x, P, masks, save=True)
        if dp:
            # ssd_loss divides by the positive count of the WHOLE batch (losses/SSDLoss.py:86): the three batch sums are
            # exchanged (24 bytes) BEFORE the gradient is scaled and back-propagated, so the N-rank step optimises
            # exactly the single-process objective on the concatenated batch
            sums, dy = hp.ssd_loss_parts(y_hat, y, 10, want_grad=True)
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            loss = hp.ssd_loss_finish(sums, dy)
        else:
            loss, dy, _ = hp.ssd_loss_fwd_bwd(y_hat, y, 10, want_grad=True)
        eng.backward(saved, dy, P, G)
        if dp:
            self._reducer.launch_tail()
            self._reducer.wait()
        self.opt.step(grads_in_flat=True)
        return loss, y_hat
