"""`SSD` with the reference's constructor, parameter names and forward signature (models/SSD.py:87-255,
models/BaseSSDModel.py:10-69).  The nn modules only hold the parameters; the arithmetic runs in the
HIP SSD engine (ssdstack.py)."""
import torch
import torch.nn as nn

from .. import hotpath as hp
from ..dataparallel import dropout_stream
from ..datasets.utils import ReduceSSDBoundingBoxes
from ..ssdstack import SSDStack, SSDStackFn, block_specs, param_names


class SeparableResidualBlock(nn.Module):
    """Parameter holder (models/SSD.py:13-64); forward lives in the engine."""

    def __init__(self, in_filters, out_filters, dropout=0.25, use_max_pool=False, bias=True):
        super().__init__()
        self.in_filters, self.out_filters = in_filters, out_filters
        if in_filters != out_filters:
            self.pointwise_conv_skip = nn.Conv2d(in_filters, out_filters, kernel_size=(1, 1), padding=0, bias=bias)
        self.conv1 = nn.Conv2d(in_filters, out_filters, kernel_size=(3, 3), padding=1, bias=bias)
        self.conv2 = nn.Conv2d(out_filters, out_filters, kernel_size=(3, 3), padding=1, bias=bias)
        self.use_max_pool = use_max_pool
        self.dropout = dropout


class SSD(nn.Module):
    def __init__(self, filters, input_shape, probability_threshold=0.5, iou_threshold=0.5, priors=None):
        super().__init__()
        if priors is not None:
            raise NotImplementedError("custom priors are not supported: they follow from the patch sizes")
        self.input_shape = input_shape
        self.probability_threshold = probability_threshold
        self.iou_threshold = iou_threshold
        self.filters = filters
        self.patch_sizes = (60, 30, 15, 7)
        _, self.height, self.width = input_shape
        self.reduce_bounding_boxes = ReduceSSDBoundingBoxes(probability_threshold=probability_threshold,
                                                            iou_threshold=iou_threshold, input_shape=input_shape,
                                                            patch_sizes=self.patch_sizes)
        self.input_normalizer = nn.Conv2d(3, filters, kernel_size=(3, 3), stride=(2, 2), padding=1, bias=True)
        specs = block_specs(filters)
        self.feature_extractor = nn.Sequential(*[SeparableResidualBlock(ci, co, use_max_pool=p) for _, ci, co, p, _ in specs[:9]])
        cont, ext = [], []
        for _, ci, co, p, _ in specs[9:]:
            cont.append(nn.Sequential(SeparableResidualBlock(ci, co, use_max_pool=p)))
            ext.append(nn.Sequential(nn.Linear(co, 5)))
        self.continue_layers = nn.ModuleList(cont)
        self.extracting_layers = nn.ModuleList(ext)
        self._engine = None
        self._injected_masks = None
        self._drop_seed = 0x55D
        self._drop_calls = 0

    @property
    def engine(self) -> SSDStack:
        if self._engine is None:
            self._engine = SSDStack(self.filters, self.input_shape[1])
        return self._engine

    def named_stack_params(self):
        names = param_names(self.filters)
        sd = dict(self.named_parameters())
        return names, [sd[n] for n in names]

    def set_dropout_masks(self, masks):
        self._injected_masks = masks

    def _draw_masks(self, n, device):
        if self._injected_masks is not None:
            return {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in self._injected_masks.items()}
        self._drop_calls += 1
        base, first = dropout_stream(self._drop_calls, n)
        specs = block_specs(self.filters)
        drawn = hp.dropout_scales_layers(n, [co for _, _, co, _, _ in specs], [0.25] * len(specs), self._drop_seed,
                                         base, first, device)
        return {spec[0]: t for spec, t in zip(specs, drawn)}

    def non_max_suppression(self, x):
        if len(x.shape) == 3:
            rows, counts = self.reduce_bounding_boxes.forward_batch(x)
            from .BaseModel import split_rows
            return split_rows(rows, counts)
        return self.reduce_bounding_boxes(x)

    def single_non_max_suppression(self, x):
        return self.reduce_bounding_boxes(x)

    def to_torchscript(self, file_path=None):
        """Scripted inference module through `fdet::ssd_forward` (torchscript.ScriptableSSD); with predict == 1 it returns
        the boxes of image 0 (TorchScript cannot return this forward's per-image tuple)."""
        from ..torchscript import to_torchscript
        return to_torchscript(self, file_path)

    def forward(self, x: torch.Tensor, predict: torch.Tensor = torch.tensor(0)):
        if predict == 1:
            if x.dim() == 3:
                x = x.unsqueeze(0)
            size = tuple(self.input_shape[1:])
            x = hp.resize_bilinear_norm(x, size) if tuple(x.shape[-2:]) != size or x.dtype == torch.uint8 else x.float() / 255.0
        if not x.is_cuda:
            raise hp.N.FdetError("the SSD stack runs on the GPU only (no CPU fallback): move model and input to cuda")
        names, params = self.named_stack_params()
        masks = self._draw_masks(x.shape[0], x.device) if self.training else None
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            y = SSDStackFn.apply(self.engine, masks, names, x, *params)
        else:
            y = self.engine.forward(x, {n: p.detach() for n, p in zip(names, params)}, masks, save=False)[0]
        if predict == 1:
            y = self.non_max_suppression(y)
        return y
