"""`MobilenetV3Backbone` with the reference's constructor, state-dict names and forward signature
(models/MobilenetV3Backbone.py:11-60).  The reference obtains the backbone from
`timm.create_model("tf_mobilenetv3_small_100", pretrained=...)`; here the same module tree (conv_stem, bn1, act1, blocks:
`feature_extractor.{0,1,3.<stage>.<block>...}`) is declared directly, so a state dict of the reference loads by name.

The modules only hold parameters.  Inference (eval mode) runs in the bf16 NHWC engine (mobilenetstack.MobileNetStack,
BatchNorm folded).  Training mode (round 4) runs mobilenet_train.MobileNetTrainEngine: fp32 NCHW, BatchNorm with batch
statistics (running statistics updated in place), hand-written backward behind an autograd bridge, so that
`ModelMeta(model=MobilenetV3Backbone(...)).training_step(batch, i)["loss"].backward(); optimizer.step()` works as it does in
the reference (models/ModelMeta.py:115-227).
"""
import warnings

import torch
import torch.nn as nn

from .. import hotpath as hp
from ..mobilenetstack import BLOCKS, BN_EPS, FEATURES, STAGES, MobileNetStack
from ..mobilenet_train import MobileNetTrainEngine, MobileNetTrainFn
from .BaseModel import BaseModel


class _EngineShim:
    """What ModelMeta / SAMSGD expect of `model.engine` (convstack.ConvStack): a hook that is called after the parameters
    were updated behind torch's back."""

    def __init__(self, model):
        self._model = model

    def mark_params_dirty(self):
        self._model._mb_key = None


def _bn(c):
    return nn.BatchNorm2d(c, eps=BN_EPS, momentum=0.01)


class _SE(nn.Module):
    def __init__(self, c, r):
        super().__init__()
        self.conv_reduce = nn.Conv2d(c, r, 1)
        self.conv_expand = nn.Conv2d(r, c, 1)


class _DepthwiseSeparable(nn.Module):
    def __init__(self, ci, co, k, se):
        super().__init__()
        self.conv_dw = nn.Conv2d(ci, ci, k, groups=ci, bias=False)
        self.bn1 = _bn(ci)
        if se:
            self.se = _SE(ci, se)
        self.conv_pw = nn.Conv2d(ci, co, 1, bias=False)
        self.bn2 = _bn(co)


class _InvertedResidual(nn.Module):
    def __init__(self, ci, ce, co, k, se):
        super().__init__()
        self.conv_pw = nn.Conv2d(ci, ce, 1, bias=False)
        self.bn1 = _bn(ce)
        self.conv_dw = nn.Conv2d(ce, ce, k, groups=ce, bias=False)
        self.bn2 = _bn(ce)
        if se:
            self.se = _SE(ce, se)
        self.conv_pwl = nn.Conv2d(ce, co, 1, bias=False)
        self.bn3 = _bn(co)


class _ConvBnAct(nn.Module):
    def __init__(self, ci, co):
        super().__init__()
        self.conv = nn.Conv2d(ci, co, 1, bias=False)
        self.bn1 = _bn(co)


def _blocks():
    stages = []
    for idx in STAGES:
        mods = []
        for b in idx:
            kind, ci, ce, co, k, s, act, se = BLOCKS[b]
            mods.append(_DepthwiseSeparable(ci, co, k, se) if kind == "ds" else _InvertedResidual(ci, ce, co, k, se))
        stages.append(nn.Sequential(*mods))
    stages.append(nn.Sequential(_ConvBnAct(96, FEATURES)))
    return nn.Sequential(*stages)


class MobilenetV3Backbone(BaseModel):
    def __init__(self, filters, input_shape, num_of_patches, probability_threshold=0.5, iou_threshold=0.5, pretrained=True,
                 input_kernel_size=10, input_stride=8, output_kernel_size=3, output_padding=0):
        super().__init__(filters, input_shape, num_of_patches=num_of_patches,
                         probability_threshold=probability_threshold, iou_threshold=iou_threshold)
        self.pretrained = pretrained
        if pretrained:
            warnings.warn("MobilenetV3Backbone(pretrained=True): the ImageNet weights are a network download in the reference "
                          "(timm); none is attempted here -- parameters are randomly initialised until load_state_dict()")
        if output_kernel_size != 3:
            raise ValueError("MobilenetV3Backbone: the HIP head kernel implements the reference's default output_kernel_size=3")
        if input_shape[1] % 32 or input_shape[2] % 32 or input_shape[1] // 32 != num_of_patches or input_shape[2] // 32 != num_of_patches:
            raise ValueError(f"MobilenetV3Backbone: the backbone has stride 32, so input {tuple(input_shape)} gives a "
                             f"{input_shape[1] // 32}x{input_shape[2] // 32} grid, not {num_of_patches}x{num_of_patches}")
        # children()[:-5] of the timm model: conv_stem, bn1, act1, blocks
        self.feature_extractor = nn.Sequential(nn.Conv2d(3, 16, 3, stride=2, bias=False), _bn(16), nn.Hardswish(), _blocks())
        self.out = nn.Conv2d(FEATURES, 5, stride=(1, 1), kernel_size=(3, 3), padding=1)
        self._mb = MobileNetStack()
        self._mb_key = None
        self._train_engine = MobileNetTrainEngine()
        self._shim = _EngineShim(self)

    @property
    def engine(self):
        return self._shim

    def head_loss_fusable(self) -> bool:
        return False

    def _packed_engine(self) -> MobileNetStack:
        sd = self.state_dict(keep_vars=True)
        key = tuple((t.data_ptr(), t._version) for t in sd.values())
        if key != self._mb_key:
            self._mb.pack({n: t.detach() for n, t in sd.items()})
            self._mb_key = key
        return self._mb

    def _train_forward(self, x: torch.Tensor) -> torch.Tensor:
        """Training-mode forward (batch statistics) with the autograd bridge to the hand-written backward."""
        if not x.is_cuda:
            raise hp.N.FdetError("the MobileNet training path runs on the GPU only (no CPU fallback): move model and input to cuda")
        params = dict(self.named_parameters())
        names = list(params.keys())
        for n, p_ in params.items():
            if p_.dtype != torch.float32 or not p_.is_contiguous():
                raise TypeError(f"MobilenetV3Backbone: parameter {n} must be contiguous float32")
        buffers = {n: b for n, b in self.named_buffers()}
        return MobileNetTrainFn.apply(self._train_engine, names, buffers, x, *[params[n] for n in names])

    def _stack_forward(self, x: torch.Tensor) -> torch.Tensor:
        """(N,3,H,W) f32 in [0,1] or uint8 -> (N,5,S,S) sigmoid maps (what BaseModel.graphed_predict captures)."""
        if self.training:
            return self._train_forward(x.float() / 255.0 if x.dtype == torch.uint8 else x)
        with torch.no_grad():
            return self._packed_engine().forward(x)

    def forward_frames(self, x: torch.Tensor) -> torch.Tensor:
        """`self(self.resize(x) / 255.0)` for frames (BaseModel.forward_frames): uint8 at the model size goes straight to the
        stem (/255 fused there), everything else through the on-device resize / normalisation."""
        if x.dim() == 3:
            x = x.unsqueeze(0)
        if not (x.dtype == torch.uint8 and tuple(x.shape[-2:]) == tuple(self.input_shape[1:])):
            x = self._preprocess(x)
        return self._stack_forward(x)

    def forward(self, x: torch.Tensor, predict: torch.Tensor = torch.tensor(0)):
        if predict == 1:
            x = self.forward_frames(x)
        elif x.dtype != torch.float32:
            raise TypeError(f"MobilenetV3Backbone.forward: expected float32 in [0,1], got {x.dtype}")
        else:
            x = self._stack_forward(x)
        if predict == 1:
            x = self.single_non_max_suppression(x[0])
        return x

    def to_torchscript(self, file_path=None):
        """Scripted inference module through `fdet::mobilenet_forward` (torchscript.ScriptableMobilenet)."""
        from ..torchscript import to_torchscript
        return to_torchscript(self, file_path)
