"""`ssd_loss` / `hard_negative_mining` with the reference's names and argument order
(losses/SSDLoss.py:25-86); mining, both loss terms and the gradient run in fdet_ssd_loss_fwd_bwd."""
import torch

from .. import hotpath as hp


class _SsdLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, ratio):
        loss, grad, _ = hp.ssd_loss_fwd_bwd(pred.detach(), target, ratio, want_grad=True)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def ssd_loss(confidence, predicted_locations, labels, gt_locations, neg_pos_ratio):
    """confidence (B,P), predicted_locations (B,P,4), labels (B,P), gt_locations (B,P,4) -> scalar.
    Called as ssd_loss(y_hat[:,:,0], y_hat[:,:,1:], y[:,:,0], y[:,:,1:], 10) (models/ModelMetaSSD.py:127-129)."""
    pred = torch.cat([confidence.unsqueeze(-1), predicted_locations], dim=-1)
    target = torch.cat([labels.unsqueeze(-1), gt_locations], dim=-1)
    return _SsdLossFn.apply(pred, target.detach(), int(neg_pos_ratio))


def hard_negative_mining(loss, labels, neg_pos_ratio):
    """(N,P) losses and labels -> bool mask of the priors that contribute (positives + the
    neg_pos_ratio * num_pos hardest negatives per image).  `loss` must be -log(confidence) as at the
    reference's only call site (SSDLoss.py:68-69): the kernel ranks on that quantity."""
    conf = torch.exp(-loss)
    pred = torch.zeros(*loss.shape, 5, device=loss.device)
    pred[..., 0] = conf
    tgt = torch.zeros_like(pred)
    tgt[..., 0] = labels
    _, _, mask = hp.ssd_loss_fwd_bwd(pred, tgt, neg_pos_ratio, want_grad=False, want_mask=True)
    return mask.bool()
