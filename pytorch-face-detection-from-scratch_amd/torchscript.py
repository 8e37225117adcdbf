"""TorchScript export through custom operators.

The reference exports its trained models with `ModelMeta.to_torchscript(path)` (train_model.py:61) and
`torch.jit.script(model)` (demo_scripts/convert_checkpoint_to_scripted_model.py:51-54); the shipped archives call
`ops.torchvision.nms`.  The mirror's forward is a sequence of HIP launches behind a C-ABI, which TorchScript cannot see
into, so the path is exposed to it as dispatcher operators (`torch.library`, namespace `fdet`):

    fdet::preprocess(Tensor x, int height, int width) -> Tensor                      resize + /255 (PoolResnet.py:91-97)
    fdet::stack_forward(Tensor x, Tensor[] params, int[] geometry) -> Tensor         the conv stack, eval mode
    fdet::reduce_bounding_boxes(Tensor maps, float pt, float iou, float w, float h) -> (Tensor, Tensor)
    fdet::nms(Tensor boxes, Tensor scores, float iou_threshold) -> Tensor            torchvision.ops.nms semantics
    fdet::mobilenet_forward(Tensor x, Tensor[] state, str[] names) -> Tensor         MobileNetV3-small backbone + head, eval mode
    fdet::ssd_forward(Tensor x, Tensor[] params, int filters, int size) -> Tensor    the SSD stack (models/SSD.py:206-255), eval mode
    fdet::ssd_reduce_bounding_boxes(Tensor y, float pt, float iou, int w, int h, int[] patch_sizes, bool with_priors) -> (Tensor, Tensor)

and `ScriptableDetector` is a small scriptable module (same parameter names as the reference: `conv1`,
`residual_blocks.k.conv{1,2}`, `out`) whose `forward(x, predict)` calls them.  A saved archive loads with
`torch.jit.load` in any process that has imported this package (the operators are registered at import).  Inference
only: the operators have no autograd formula (training goes through models.ModelMeta).

`torchvision::nms` is registered too when torchvision is absent, so scripted code written against the reference
(`torchvision.ops.nms(boxes, scores, iou_threshold)`) resolves to the HIP kernel.
"""
from __future__ import annotations

import importlib.util
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import hotpath as hp
from .convstack import ConvStack, StackGeometry, param_names

_KINDS = ("poolresnet", "resnet")
_engines: Dict[Tuple[int, ...], ConvStack] = {}
_lib = None
_tv_lib = None


def _geometry_list(g: StackGeometry) -> List[int]:
    return [_KINDS.index(g.kind), g.filters, g.in_ch, g.H, g.W, g.S, g.num_blocks, g.stem_k, g.stem_s, g.stem_p,
            g.head_k, g.head_p, g.pool_mult]


def _engine_for(geo: List[int]) -> ConvStack:
    key = tuple(int(v) for v in geo)
    if len(key) != 13:
        raise ValueError(f"fdet::stack_forward: geometry must hold 13 integers, got {len(key)}")
    eng = _engines.get(key)
    if eng is None:
        eng = ConvStack(StackGeometry(_KINDS[key[0]], *key[1:]))
        _engines[key] = eng
    return eng


def _op_stack_forward(x: torch.Tensor, params: List[torch.Tensor], geo: List[int]) -> torch.Tensor:
    eng = _engine_for(geo)
    names = param_names(eng.geo.num_blocks)
    if len(params) != len(names):
        raise ValueError(f"fdet::stack_forward: expected {len(names)} parameter tensors, got {len(params)}")
    P = {n: p.detach() for n, p in zip(names, params)}
    # the engine recognises "same weights" by (address, version): keep the tensors it packed from ALIVE, so that a module
    # loaded later cannot be handed the same addresses by the caching allocator and be served the old packed panels
    eng._src_refs = list(params)
    return eng.forward(x.detach(), P, None, save=False)[0]


_mb_engines: Dict[tuple, object] = {}


def _op_mobilenet_forward(x: torch.Tensor, state: List[torch.Tensor], names: List[str]) -> torch.Tensor:
    """MobileNetV3-small backbone + head (models/MobilenetV3Backbone.py:49-60), eval mode: `state` / `names` are the
    module's state_dict (BatchNorm running statistics included); the packed bf16 engine is cached per set of tensors."""
    from .mobilenetstack import MobileNetStack
    if len(state) != len(names):
        raise ValueError("fdet::mobilenet_forward: state and names differ in length")
    key = tuple((t.data_ptr(), t._version) for t in state)
    eng = _mb_engines.get(key)
    if eng is None:
        if len(_mb_engines) > 8:
            _mb_engines.clear()
        eng = MobileNetStack()
        eng.pack({n: t.detach() for n, t in zip(names, state)})
        eng._src_refs = list(state)                          # strong references: a cached key's addresses cannot be recycled
        _mb_engines[key] = eng
    return eng.forward(x.detach())


_ssd_engines: Dict[Tuple[int, int], object] = {}


def _op_ssd_forward(x: torch.Tensor, params: List[torch.Tensor], filters: int, size: int) -> torch.Tensor:
    """`SSD.forward` up to the prior application (models/SSD.py:206-255), eval mode: (N,3,size,size) f32 -> (N,4774,5)."""
    from .ssdstack import SSDStack, param_names as ssd_param_names
    eng = _ssd_engines.get((filters, size))
    if eng is None:
        eng = _ssd_engines[(filters, size)] = SSDStack(filters, size)
    names = ssd_param_names(filters)
    if len(params) != len(names):
        raise ValueError(f"fdet::ssd_forward: expected {len(names)} parameter tensors, got {len(params)}")
    eng._src_refs = list(params)                             # as in _op_stack_forward
    return eng.forward(x.detach(), {n: p.detach() for n, p in zip(names, params)}, None, save=False)[0]


def _op_ssd_reduce(y: torch.Tensor, pt: float, iou: float, w: int, h: int, patch_sizes: List[int], with_priors: bool):
    rows, counts = hp.ssd_reduce_bounding_boxes(y, pt, iou, w, h, tuple(patch_sizes), with_priors)
    return rows, counts.to(torch.int64)


def _op_preprocess(x: torch.Tensor, height: int, width: int) -> torch.Tensor:
    if x.dim() == 3:
        x = x.unsqueeze(0)
    if tuple(x.shape[-2:]) != (height, width):
        return hp.resize_bilinear_norm(x, (height, width))
    if x.dtype == torch.uint8:
        return hp.u8_to_f32_norm(x)
    return x.float() / 255.0


def _op_reduce(maps: torch.Tensor, pt: float, iou: float, w: float, h: float):
    rows, counts = hp.reduce_bounding_boxes(maps, pt, iou, w, h)
    return rows, counts.to(torch.int64)


def _op_nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    return hp.nms(boxes, scores, iou_threshold)


def _no_cpu(name):
    def fn(*args, **kwargs):
        raise hp.N.FdetError(f"fdet::{name} runs on the GPU only (no CPU fallback): move the model and its inputs to cuda")
    return fn


def register_ops() -> None:
    """Idempotent: defines the fdet:: operators (and torchvision::nms when torchvision is not installed)."""
    global _lib, _tv_lib
    if _lib is not None:
        return
    lib = torch.library.Library("fdet", "DEF")
    lib.define("preprocess(Tensor x, int height, int width) -> Tensor")
    lib.define("stack_forward(Tensor x, Tensor[] params, int[] geometry) -> Tensor")
    lib.define("reduce_bounding_boxes(Tensor maps, float pt, float iou, float w, float h) -> (Tensor, Tensor)")
    lib.define("nms(Tensor boxes, Tensor scores, float iou_threshold) -> Tensor")
    lib.define("mobilenet_forward(Tensor x, Tensor[] state, str[] names) -> Tensor")
    lib.define("ssd_forward(Tensor x, Tensor[] params, int filters, int size) -> Tensor")
    lib.define("ssd_reduce_bounding_boxes(Tensor y, float pt, float iou, int w, int h, int[] patch_sizes, bool with_priors) -> (Tensor, Tensor)")
    for name, fn in (("preprocess", _op_preprocess), ("stack_forward", _op_stack_forward), ("mobilenet_forward", _op_mobilenet_forward), ("ssd_forward", _op_ssd_forward),
                     ("ssd_reduce_bounding_boxes", _op_ssd_reduce),
                     ("reduce_bounding_boxes", _op_reduce), ("nms", _op_nms)):
        lib.impl(name, fn, "CUDA")
        lib.impl(name, _no_cpu(name), "CPU")
    _lib = lib
    if importlib.util.find_spec("torchvision") is None:
        try:
            tv = torch.library.Library("torchvision", "DEF")
            tv.define("nms(Tensor dets, Tensor scores, float iou_threshold) -> Tensor")
            tv.impl("nms", _op_nms, "CUDA")
            tv.impl("nms", _no_cpu("nms"), "CPU")
            _tv_lib = tv
        except RuntimeError:                             # somebody else defined it first: leave theirs alone
            _tv_lib = None


register_ops()


class _BlockParams(nn.Module):
    """conv1 / conv2 parameter holder with the reference's names (models/PoolResnet.py:11-31)."""

    def __init__(self, conv1: nn.Conv2d, conv2: nn.Conv2d):
        super().__init__()
        self.conv1 = conv1
        self.conv2 = conv2

    def forward(self, x: torch.Tensor) -> torch.Tensor:      # never called: the arithmetic is fdet::stack_forward
        return x


class ScriptableDetector(nn.Module):
    """`forward(x, predict=tensor(0))` of the reference models (PoolResnet.py:93-105, Resnet.py:87-99) in eval mode,
    written so that `torch.jit.script` accepts it.  Shares its parameters with the model it was built from."""

    def __init__(self, model):
        super().__init__()
        g = model._geometry()
        self.geometry: List[int] = _geometry_list(g)
        self.height: int = int(g.H)
        self.width: int = int(g.W)
        self.num_of_patches: int = int(g.S)
        self.probability_threshold: float = float(model.reduce_bounding_boxes.probability_threshold)
        self.iou_threshold: float = float(model.reduce_bounding_boxes.iou_threshold)
        self.conv1 = model.conv1
        self.residual_blocks = nn.Sequential(*[_BlockParams(b.conv1, b.conv2) for b in model.residual_blocks])
        self.out = model.out

    def forward(self, x: torch.Tensor, predict: torch.Tensor = torch.tensor(0)) -> torch.Tensor:
        want_boxes = bool(predict == 1)
        if want_boxes:
            x = torch.ops.fdet.preprocess(x, self.height, self.width)
        params: List[torch.Tensor] = [self.conv1.weight, torch.jit._unwrap_optional(self.conv1.bias)]
        for blk in self.residual_blocks:
            params.append(blk.conv1.weight)
            params.append(torch.jit._unwrap_optional(blk.conv1.bias))
            params.append(blk.conv2.weight)
            params.append(torch.jit._unwrap_optional(blk.conv2.bias))
        params.append(self.out.weight)
        params.append(torch.jit._unwrap_optional(self.out.bias))
        y = torch.ops.fdet.stack_forward(x, params, self.geometry)
        if want_boxes:
            # boxes of image 0 only, as the reference (PoolResnet.py:103-104); (0,5) when nothing passes the threshold
            rows, counts = torch.ops.fdet.reduce_bounding_boxes(y[0:1], self.probability_threshold, self.iou_threshold,
                                                                float(self.width), float(self.height))
            k = int(counts[0])
            y = rows[0, :k]
        return y


class ScriptableMobilenet(nn.Module):
    """`forward(x, predict=tensor(0))` of models.MobilenetV3Backbone in eval mode for `torch.jit.script`.  The module's
    state (weights and BatchNorm statistics) travels as a tensor-list attribute next to its state_dict names; the
    arithmetic is `fdet::mobilenet_forward`.  Inference only, like the eager mirror."""

    def __init__(self, model):
        super().__init__()
        sd = model.state_dict()
        self.names: List[str] = list(sd.keys())
        self.state: List[torch.Tensor] = [t.detach().clone() for t in sd.values()]
        self.height: int = int(model.input_shape[1])
        self.width: int = int(model.input_shape[2])
        self.probability_threshold: float = float(model.reduce_bounding_boxes.probability_threshold)
        self.iou_threshold: float = float(model.reduce_bounding_boxes.iou_threshold)

    def forward(self, x: torch.Tensor, predict: torch.Tensor = torch.tensor(0)) -> torch.Tensor:
        want_boxes = bool(predict == 1)
        if want_boxes:
            if x.dim() == 3:
                x = x.unsqueeze(0)
            # uint8 frames at the model size go straight to the stem (/255 fused there), as in the eager model
            if not (x.dtype == torch.uint8 and x.size(-2) == self.height and x.size(-1) == self.width):
                x = torch.ops.fdet.preprocess(x, self.height, self.width)
        y = torch.ops.fdet.mobilenet_forward(x, self.state, self.names)
        if want_boxes:
            rows, counts = torch.ops.fdet.reduce_bounding_boxes(y[0:1], self.probability_threshold, self.iou_threshold,
                                                                float(self.width), float(self.height))
            k = int(counts[0])
            y = rows[0, :k]
        return y


class ScriptableSSD(nn.Module):
    """`forward(x, predict=tensor(0))` of models.SSD.SSD (models/SSD.py:206-255) in eval mode for `torch.jit.script`.  With
    predict == 1 the reference returns a TUPLE of per-image results, which TorchScript cannot mix with the tensor of the
    other branch: the scripted module returns the boxes of image 0, like the YOLO models' scripted forward."""

    def __init__(self, model):
        super().__init__()
        names, params = model.named_stack_params()
        self.params_list: List[torch.Tensor] = [p.detach().clone() for p in params]
        self.filters: int = int(model.filters)
        self.size: int = int(model.input_shape[1])
        self.patch_sizes: List[int] = [int(v) for v in model.patch_sizes]
        self.rw: int = int(model.reduce_bounding_boxes.width)
        self.rh: int = int(model.reduce_bounding_boxes.height)
        self.probability_threshold: float = float(model.reduce_bounding_boxes.probability_threshold)
        self.iou_threshold: float = float(model.reduce_bounding_boxes.iou_threshold)
        self.with_priors: bool = bool(model.reduce_bounding_boxes.with_priors)

    def forward(self, x: torch.Tensor, predict: torch.Tensor = torch.tensor(0)) -> torch.Tensor:
        want_boxes = bool(predict == 1)
        if want_boxes:
            x = torch.ops.fdet.preprocess(x, self.size, self.size)
        y = torch.ops.fdet.ssd_forward(x, self.params_list, self.filters, self.size)
        if want_boxes:
            rows, counts = torch.ops.fdet.ssd_reduce_bounding_boxes(y[0:1], self.probability_threshold, self.iou_threshold,
                                                                    self.rw, self.rh, self.patch_sizes, self.with_priors)
            k = int(counts[0])
            y = rows[0, :k]
        return y


def to_torchscript(model, file_path=None) -> torch.jit.ScriptModule:
    """`torch.jit.script` of the model's inference path; saved to `file_path` when given (Lightning's
    `LightningModule.to_torchscript(file_path)` contract, train_model.py:61)."""
    from .models.MobilenetV3Backbone import MobilenetV3Backbone
    from .models.SSD import SSD
    wrapper = ScriptableMobilenet(model) if isinstance(model, MobilenetV3Backbone) else (ScriptableSSD(model) if isinstance(model, SSD) else ScriptableDetector(model))
    scripted = torch.jit.script(wrapper.eval())
    if file_path is not None:
        torch.jit.save(scripted, str(file_path))
    return scripted
