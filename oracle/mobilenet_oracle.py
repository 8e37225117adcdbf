"""CPU oracle of the MobileNetV3-small backbone model (BASELINE.json config 5).  TEST
INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: models/MobilenetV3Backbone.py:11-60 -- `timm.create_model("tf_mobilenetv3_small_100")` without its
classifier (children()[:-5]: conv_stem, bn1, act1, blocks) + `Conv2d(576, 5, 3, padding=1)` + sigmoid.

PARITY UNPINNED.  timm is not installed here and cannot be fetched, and the shipped TorchScript archive
(saved_models/official/MobilenetV3Backbone/medium_model_15x15_480.pth) may not be executed, so no output of the
reference exists to check this restatement against.  What IS taken from the reference: the parameter tensors of that
archive (raw storage bytes, tools/make_goldens_mobilenet.py) and, from its embedded source text (read, not run) and
the tensor shapes, the architecture:

  conv_stem Conv2dSame(3,16,3,s2) - BN(eps 1e-3) - Hardswish
  stage 0: DepthwiseSeparable  dw3x3 s2 (SAME) - BN - ReLU - SE(16->8->16) - pw 16->16 - BN            (no skip: stride 2)
  stage 1: InvertedResidual    16->72->24  k3 s2 ReLU ; 24->88->24 k3 s1 ReLU (+skip)
  stage 2: InvertedResidual+SE 24->96->40  k5 s2 Hardswish ; 2 x (40->240->40 k5 s1 Hardswish, +skip)
  stage 3: InvertedResidual+SE 40->120->48 k5 s1 ; 48->144->48 k5 s1 (+skip)
  stage 4: InvertedResidual+SE 48->288->96 k5 s2 ; 2 x (96->576->96 k5 s1, +skip)
  stage 5: ConvBnAct           96->576 1x1 - BN - Hardswish
  out Conv2d(576,5,3,p1) - sigmoid
  InvertedResidual: pw - BN - act - dw (stride 2: TF "SAME" asymmetric padding; stride 1: k//2) - BN - act - [SE] - pwl - BN
  SqueezeExcite: x * hardsigmoid(expand(relu(reduce(mean_hw(x)))))   (1x1 convs with bias)
"""
from typing import Dict, List

import torch
import torch.nn.functional as F

BN_EPS = 1e-3

# (kind, cin, cexp, cout, kernel, stride, act, se_reduce or 0)
BLOCKS = [
    ("ds", 16, 16, 16, 3, 2, "relu", 8),
    ("ir", 16, 72, 24, 3, 2, "relu", 0),
    ("ir", 24, 88, 24, 3, 1, "relu", 0),
    ("ir", 24, 96, 40, 5, 2, "hswish", 24),
    ("ir", 40, 240, 40, 5, 1, "hswish", 64),
    ("ir", 40, 240, 40, 5, 1, "hswish", 64),
    ("ir", 40, 120, 48, 5, 1, "hswish", 32),
    ("ir", 48, 144, 48, 5, 1, "hswish", 40),
    ("ir", 48, 288, 96, 5, 2, "hswish", 72),
    ("ir", 96, 576, 96, 5, 1, "hswish", 144),
    ("ir", 96, 576, 96, 5, 1, "hswish", 144),
]
STAGE_OF_BLOCK = [0, 1, 1, 2, 2, 2, 3, 3, 4, 4, 4]
INDEX_IN_STAGE = [0, 0, 1, 0, 1, 2, 0, 1, 0, 1, 2]


def param_names() -> List[str]:
    """state_dict order of the reference module (= storage order of the archive's 242 tensors)."""
    def bn(p):
        return [p + ".weight", p + ".bias", p + ".running_mean", p + ".running_var", p + ".num_batches_tracked"]
    names = ["feature_extractor.0.weight"] + bn("feature_extractor.1")
    for b, (kind, ci, ce, co, k, s, act, se) in enumerate(BLOCKS):
        p = f"feature_extractor.3.{STAGE_OF_BLOCK[b]}.{INDEX_IN_STAGE[b]}"
        if kind == "ds":
            names += [p + ".conv_dw.weight"] + bn(p + ".bn1")
            names += [p + ".se.conv_reduce.weight", p + ".se.conv_reduce.bias", p + ".se.conv_expand.weight", p + ".se.conv_expand.bias"]
            names += [p + ".conv_pw.weight"] + bn(p + ".bn2")
        else:
            names += [p + ".conv_pw.weight"] + bn(p + ".bn1") + [p + ".conv_dw.weight"] + bn(p + ".bn2")
            if se:
                names += [p + ".se.conv_reduce.weight", p + ".se.conv_reduce.bias", p + ".se.conv_expand.weight", p + ".se.conv_expand.bias"]
            names += [p + ".conv_pwl.weight"] + bn(p + ".bn3")
    names += ["feature_extractor.3.5.0.conv.weight"] + bn("feature_extractor.3.5.0.bn1")
    names += ["out.weight", "out.bias"]
    return names


def param_shapes() -> Dict[str, tuple]:
    sh = {"feature_extractor.0.weight": (16, 3, 3, 3), "out.weight": (5, 576, 3, 3), "out.bias": (5,)}

    def bn(p, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            sh[p + "." + s] = (c,)
        sh[p + ".num_batches_tracked"] = ()
    bn("feature_extractor.1", 16)
    for b, (kind, ci, ce, co, k, s, act, se) in enumerate(BLOCKS):
        p = f"feature_extractor.3.{STAGE_OF_BLOCK[b]}.{INDEX_IN_STAGE[b]}"
        if kind == "ds":
            sh[p + ".conv_dw.weight"] = (ci, 1, k, k); bn(p + ".bn1", ci)
            sh[p + ".conv_pw.weight"] = (co, ci, 1, 1); bn(p + ".bn2", co)
            cse = ci
        else:
            sh[p + ".conv_pw.weight"] = (ce, ci, 1, 1); bn(p + ".bn1", ce)
            sh[p + ".conv_dw.weight"] = (ce, 1, k, k); bn(p + ".bn2", ce)
            sh[p + ".conv_pwl.weight"] = (co, ce, 1, 1); bn(p + ".bn3", co)
            cse = ce
        if se:
            sh[p + ".se.conv_reduce.weight"] = (se, cse, 1, 1); sh[p + ".se.conv_reduce.bias"] = (se,)
            sh[p + ".se.conv_expand.weight"] = (cse, se, 1, 1); sh[p + ".se.conv_expand.bias"] = (cse,)
    sh["feature_extractor.3.5.0.conv.weight"] = (576, 96, 1, 1); bn("feature_extractor.3.5.0.bn1", 576)
    return sh


def init_params(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random parameters with trained-network-like statistics (tests that must not depend on the archive)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for n, s in param_shapes().items():
        if n.endswith("num_batches_tracked"):
            P[n] = torch.tensor(1000, dtype=torch.int64)
        elif n.endswith("running_var"):
            P[n] = torch.rand(s, generator=g) * 0.9 + 0.3
        elif n.endswith("running_mean"):
            P[n] = torch.randn(s, generator=g) * 0.2
        elif ".bn" in n or n.startswith("feature_extractor.1.") or n.endswith("bn1.bias") or n.endswith("bn1.weight"):
            P[n] = (torch.rand(s, generator=g) * 0.8 + 0.6) if n.endswith("weight") else torch.randn(s, generator=g) * 0.1
        elif n.endswith("bias"):
            P[n] = torch.randn(s, generator=g) * 0.1
        else:
            fan_in = s[1] * s[2] * s[3]
            P[n] = torch.randn(s, generator=g) * (1.4 / fan_in ** 0.5)
    return P


def _same_pad(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    """timm pad_same (TF "SAME"): total = max((ceil(i/s)-1)*s + k - i, 0), split floor / ceil (extra on the right/bottom)."""
    ih, iw = x.shape[-2:]
    ph = max((-(-ih // s) - 1) * s + k - ih, 0)
    pw = max((-(-iw // s) - 1) * s + k - iw, 0)
    return F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])


def _bn(x, P, p):
    return F.batch_norm(x, P[p + ".running_mean"], P[p + ".running_var"], P[p + ".weight"], P[p + ".bias"], False, 0.1, BN_EPS)


def _act(x, kind):
    return F.relu(x) if kind == "relu" else F.hardswish(x)


def _dw(x, w, k, s):
    if s == 1:
        return F.conv2d(x, w, None, 1, k // 2, 1, x.shape[1])
    return F.conv2d(_same_pad(x, k, s), w, None, s, 0, 1, x.shape[1])


def _se(x, P, p):
    s = x.mean((2, 3), keepdim=True)
    s = F.relu(F.conv2d(s, P[p + ".conv_reduce.weight"], P[p + ".conv_reduce.bias"]))
    s = F.conv2d(s, P[p + ".conv_expand.weight"], P[p + ".conv_expand.bias"])
    return x * F.hardsigmoid(s)


def stem_forward(P: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    h = F.conv2d(_same_pad(x, 3, 2), P["feature_extractor.0.weight"], None, 2)
    return F.hardswish(_bn(h, P, "feature_extractor.1"))


def block_forward(P: Dict[str, torch.Tensor], b: int, h: torch.Tensor) -> torch.Tensor:
    """Block `b` of BLOCKS (timm DepthwiseSeparableConv / InvertedResidual forward, eval mode, drop_path off)."""
    kind, ci, ce, co, k, s, act, se = BLOCKS[b]
    p = f"feature_extractor.3.{STAGE_OF_BLOCK[b]}.{INDEX_IN_STAGE[b]}"
    skip = h
    if kind == "ds":
        h = _act(_bn(_dw(h, P[p + ".conv_dw.weight"], k, s), P, p + ".bn1"), act)
        if se:
            h = _se(h, P, p + ".se")
        h = _bn(F.conv2d(h, P[p + ".conv_pw.weight"]), P, p + ".bn2")
    else:
        h = _act(_bn(F.conv2d(h, P[p + ".conv_pw.weight"]), P, p + ".bn1"), act)
        h = _act(_bn(_dw(h, P[p + ".conv_dw.weight"], k, s), P, p + ".bn2"), act)
        if se:
            h = _se(h, P, p + ".se")
        h = _bn(F.conv2d(h, P[p + ".conv_pwl.weight"]), P, p + ".bn3")
    if s == 1 and ci == co:
        h = h + skip
    return h


def final_forward(P: Dict[str, torch.Tensor], h: torch.Tensor) -> torch.Tensor:
    h = F.conv2d(h, P["feature_extractor.3.5.0.conv.weight"])
    return F.hardswish(_bn(h, P, "feature_extractor.3.5.0.bn1"))


def features(P: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """(N,3,H,W) f32 in [0,1] -> (N,576,H/32,W/32) backbone features."""
    h = stem_forward(P, x)
    for b in range(len(BLOCKS)):
        h = block_forward(P, b, h)
    return final_forward(P, h)


def model_forward(P: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """MobilenetV3Backbone.forward(x, predict=0): (N,3,H,W) -> (N,5,H/32,W/32) (models/MobilenetV3Backbone.py:49-60)."""
    return torch.sigmoid(F.conv2d(features(P, x), P["out.weight"], P["out.bias"], padding=1))


# ---- training mode (round 4): the same graph with BatchNorm on BATCH statistics (F.batch_norm(training=True) also updates the
# running statistics in place, momentum 0.01 = timm's tf_ default), differentiable by torch autograd
BN_MOMENTUM = 0.01


def _bn_t(x, P, p):
    return F.batch_norm(x, P[p + ".running_mean"], P[p + ".running_var"], P[p + ".weight"], P[p + ".bias"], True, BN_MOMENTUM, BN_EPS)


def model_forward_train(P: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """MobilenetV3Backbone.forward in train() mode: (N,3,H,W) -> (N,5,H/32,W/32); P's running statistics are updated."""
    h = F.conv2d(_same_pad(x, 3, 2), P["feature_extractor.0.weight"], None, 2)
    h = F.hardswish(_bn_t(h, P, "feature_extractor.1"))
    for b, (kind, ci, ce, co, k, s, act, se) in enumerate(BLOCKS):
        p = f"feature_extractor.3.{STAGE_OF_BLOCK[b]}.{INDEX_IN_STAGE[b]}"
        skip = h
        if kind == "ds":
            h = _act(_bn_t(_dw(h, P[p + ".conv_dw.weight"], k, s), P, p + ".bn1"), act)
            if se:
                h = _se(h, P, p + ".se")
            h = _bn_t(F.conv2d(h, P[p + ".conv_pw.weight"]), P, p + ".bn2")
        else:
            h = _act(_bn_t(F.conv2d(h, P[p + ".conv_pw.weight"]), P, p + ".bn1"), act)
            h = _act(_bn_t(_dw(h, P[p + ".conv_dw.weight"], k, s), P, p + ".bn2"), act)
            if se:
                h = _se(h, P, p + ".se")
            h = _bn_t(F.conv2d(h, P[p + ".conv_pwl.weight"]), P, p + ".bn3")
        if s == 1 and ci == co:
            h = h + skip
    h = F.hardswish(_bn_t(F.conv2d(h, P["feature_extractor.3.5.0.conv.weight"]), P, "feature_extractor.3.5.0.bn1"))
    return torch.sigmoid(F.conv2d(h, P["out.weight"], P["out.bias"], padding=1))
