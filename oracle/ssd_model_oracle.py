"""CPU restatement of the reference's SSD model (models/SSD.py:13-255).  TEST INFRASTRUCTURE ONLY.

  SeparableResidualBlock.forward  models/SSD.py:66-84   (1x1 skip conv iff in != out; conv-lrelu-conv-lrelu-
                                                        dropout2d(0.25)-skip add-[maxpool 2, floor])
  SSD.__init__ / forward          models/SSD.py:87-255  (stem k3 s2, 9 + 4 blocks, Linear heads on NHWC,
                                                        sigmoid on the scores, apply_priors)
Pinned by tests/golden/g10_ssd_model.npz (tools/make_goldens_ssd.py runs the reference class; parameters are
regenerated from the seed: `init_params` creates the layers in the reference's construction order, and the
generator asserts that this reproduces the reference's state_dict bit for bit)."""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ssd_oracle import PATCH_SIZES, ssd_priors, ssd_loss


def block_specs(filters: int) -> Tuple[List[Tuple[str, int, int, bool]], List[int]]:
    """[(name, in, out, pool)] in forward order and the head input widths (models/SSD.py:135-186)."""
    f = filters
    fe = [(f, 2 * f, True), (2 * f, 2 * f, True)] + [(2 * f, 2 * f, False)] * 6 + [(2 * f, 4 * f, False)]
    specs = [(f"feature_extractor.{k}", i, o, p) for k, (i, o, p) in enumerate(fe)]
    heads = []
    mx = 16 * f
    for i in range(len(PATCH_SIZES)):
        cin = min(4 * f * (2 ** i), mx)
        cout = min(2 * cin, mx)
        specs.append((f"continue_layers.{i}.0", cin, cout, i != 0))
        heads.append(cout)
    return specs, heads


def init_params(filters: int, seed: int) -> Dict[str, torch.Tensor]:
    """Default torch init under torch.manual_seed(seed), layers created in the order of SSD.__init__."""
    torch.manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}

    def conv(name, ci, co, k):
        m = nn.Conv2d(ci, co, kernel_size=(k, k), padding=k // 2, bias=True)
        P[name + ".weight"] = m.weight.detach().clone(); P[name + ".bias"] = m.bias.detach().clone()

    def block(name, ci, co):
        if ci != co:
            conv(name + ".pointwise_conv_skip", ci, co, 1)
        conv(name + ".conv1", ci, co, 3)
        conv(name + ".conv2", co, co, 3)

    conv("input_normalizer", 3, filters, 3)
    specs, heads = block_specs(filters)
    for name, ci, co, _ in specs[:9]:
        block(name, ci, co)
    for i in range(len(PATCH_SIZES)):
        name, ci, co, _ = specs[9 + i]
        block(name, ci, co)
        lin = nn.Linear(co, 5)
        P[f"extracting_layers.{i}.0.weight"] = lin.weight.detach().clone()
        P[f"extracting_layers.{i}.0.bias"] = lin.bias.detach().clone()
    return P


def make_dropout_masks(filters: int, batch: int, seed: int, p: float = 0.25) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    specs, _ = block_specs(filters)
    return {name: (torch.rand(batch, co, generator=g) >= p).float() / (1.0 - p) for name, _, co, _ in specs}


def model_forward(filters: int, P: Dict[str, torch.Tensor], x: torch.Tensor,
                  masks: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """x (B,3,480,480) in [0,1] -> (B,4774,5) [sigmoid score, prior-decoded x, y, w, h] (SSD.forward, predict=0)."""
    bs = x.size(0)
    specs, _ = block_specs(filters)
    h = F.conv2d(x, P["input_normalizer.weight"], P["input_normalizer.bias"], stride=2, padding=1)
    scores, bbxs = [], []

    def block(name, ci, co, pool, t):
        skip = t if ci == co else F.conv2d(t, P[name + ".pointwise_conv_skip.weight"], P[name + ".pointwise_conv_skip.bias"])
        a = F.leaky_relu(F.conv2d(t, P[name + ".conv1.weight"], P[name + ".conv1.bias"], padding=1), 0.2)
        c = F.leaky_relu(F.conv2d(a, P[name + ".conv2.weight"], P[name + ".conv2.bias"], padding=1), 0.2)
        if masks is not None:
            c = c * masks[name][:, :, None, None]
        e = c + skip
        return F.max_pool2d(e, 2) if pool else e

    for name, ci, co, pool in specs[:9]:
        h = block(name, ci, co, pool, h)
    for i in range(len(PATCH_SIZES)):
        name, ci, co, pool = specs[9 + i]
        h = block(name, ci, co, pool, h)
        z = F.linear(h.permute(0, 2, 3, 1).contiguous(), P[f"extracting_layers.{i}.0.weight"], P[f"extracting_layers.{i}.0.bias"])
        z = z.reshape(bs, -1, 5)
        scores.append(z[..., :1]); bbxs.append(z[..., 1:5])
    y = torch.cat([torch.sigmoid(torch.cat(scores, dim=1)), torch.cat(bbxs, dim=1)], dim=2)
    mult, priors = ssd_priors(PATCH_SIZES)                          # apply_priors (models/SSD.py:206-218)
    y = y.clone().float()
    y[..., 1:2] = y[..., 1:2] * mult.repeat(repeats=(bs, 1, 1))
    y[..., 2:3] = y[..., 2:3] * mult.repeat(repeats=(bs, 1, 1))
    y[..., 1:5] = y[..., 1:5] + priors.repeat(repeats=(bs, 1, 1))
    return y


def loss_and_grads(filters, P, x, target, masks=None, neg_pos_ratio=10):
    """ssd_loss on the model output and its gradients w.r.t. every parameter (autograd)."""
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    y = model_forward(filters, Pg, x, masks)
    loss = ssd_loss(y[:, :, 0], y[:, :, 1:], target[:, :, 0], target[:, :, 1:], neg_pos_ratio)
    names = list(Pg)
    grads = torch.autograd.grad(loss, [Pg[n] for n in names])
    return loss.detach(), y.detach(), dict(zip(names, grads))
