"""`yolo_loss(pred_fm, gt_fm)` with the reference's signature (losses/YoloLoss.py:4-44),
computed by the fused HIP loss kernel (forward + analytic backward in one launch).

`yolo_loss_batch(y_hat, y)` is the batch form ModelMeta.step uses: it equals the reference's
Python loop `sum(yolo_loss(y_hat[i], y[i]))` (models/ModelMeta.py:173-176) in one launch.
"""
import torch

from .. import hotpath as hp


class _YoloLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        lpi, lsum, grad = hp.yolo_loss_fwd_bwd(pred.detach(), gt.detach(), want_grad=True)
        ctx.save_for_backward(grad)
        ctx.per_image = lpi
        return lsum.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def yolo_loss_batch(y_hat: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """(B,5,S,S),(B,5,S,S) -> 0-d tensor = sum over images of the per-image loss."""
    return _YoloLossFn.apply(y_hat, y)


def yolo_loss(pred_fm: torch.Tensor, gt_fm: torch.Tensor) -> torch.Tensor:
    """(5,S,S),(5,S,S) -> 0-d loss tensor (differentiable w.r.t. pred_fm)."""
    if pred_fm.dim() != 3 or pred_fm.shape[0] != 5:
        raise ValueError(f"yolo_loss expects (5,S,S) maps, got {tuple(pred_fm.shape)}")
    return _YoloLossFn.apply(pred_fm.unsqueeze(0), gt_fm.unsqueeze(0))
