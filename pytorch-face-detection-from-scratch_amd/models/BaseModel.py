"""`BaseModel` with the reference's constructor and methods (models/BaseModel.py:11-71)."""
import torch
import torch.nn as nn

from .. import hotpath as hp
from ..convstack import ConvStack, ConvStackFn, StackGeometry, param_names
from ..dataparallel import dropout_stream
from ..datasets.utils import ReduceBoundingBoxes


class BaseModel(nn.Module):
    def __init__(self, filters, input_shape, num_of_patches, probability_threshold=0.5, iou_threshold=0.5):
        super().__init__()
        self.input_shape = input_shape
        self.num_of_patches = num_of_patches
        assert (
            input_shape[1] % num_of_patches == 0 and input_shape[2] % num_of_patches == 0
        ), f"Input shape {input_shape} cannot be divided into {num_of_patches} patches"
        self.filters = filters
        self.probability_threshold = probability_threshold
        self.iou_threshold = iou_threshold
        self.reduce_bounding_boxes = ReduceBoundingBoxes(
            probability_threshold=probability_threshold, iou_threshold=iou_threshold,
            input_shape=self.input_shape, num_of_patches=self.num_of_patches)
        self._engine = None
        self._injected_masks = None
        self._drop_seed = 0x5EED
        self._drop_calls = 0

    # -------------------------------------------------------------- conv stack plumbing
    def _geometry(self) -> StackGeometry:
        raise NotImplementedError

    @property
    def engine(self) -> ConvStack:
        if self._engine is None:
            self._engine = ConvStack(self._geometry())
        return self._engine

    def named_stack_params(self):
        names = param_names(len(self.residual_blocks))
        sd = dict(self.named_parameters())
        return names, [sd[n] for n in names]

    def set_dropout_masks(self, masks):
        """Inject per-(n,c) Dropout2d scale factors (parity tests); None = draw on device."""
        self._injected_masks = masks

    def _draw_masks(self, n: int, device):
        if self._injected_masks is not None:
            return {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in self._injected_masks.items()}
        # one launch for all layers; the counter is (call, GLOBAL image, layer, channel): call ranges never overlap
        # whatever the batch size, and a data-parallel rank draws the planes of ITS images of the global batch
        nb, F_ = len(self.residual_blocks), self.filters
        self._drop_calls += 1
        base, first = dropout_stream(self._drop_calls, n)
        drawn = hp.dropout_scales_layers(n, [F_] * (nb + 1), [0.25] * nb + [0.5], self._drop_seed, base, first, device)
        masks = {f"residual_blocks.{k}": drawn[k] for k in range(nb)}       # ResidualBlock dropout
        masks["head"] = drawn[nb]                                          # model-level dropout
        return masks

    def _stack_forward(self, x: torch.Tensor) -> torch.Tensor:
        names, params = self.named_stack_params()
        if not x.is_cuda:
            raise hp.N.FdetError("the conv stack runs on the GPU only (no CPU fallback): move model and input to cuda")
        masks = self._draw_masks(x.shape[0], x.device) if self.training else None
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return ConvStackFn.apply(self.engine, masks, names, x, *params)
        P = {n: p.detach() for n, p in zip(names, params)}
        return self.engine.forward(x, P, masks, save=False)[0]

    def _preprocess(self, x: torch.Tensor) -> torch.Tensor:
        """`self.resize(x) / 255.0` (PoolResnet.py:91,95; BaseModel.py:64-65).  At the model
        resolution the Resize is an identity round trip and only the /255 runs;
        other sizes go through the on-device bilinear resize (uint8 frames stay 1 byte per pixel
        on the way to the GPU)."""
        if x.dim() == 3:
            x = x.unsqueeze(0)
        if not x.is_cuda:
            raise hp.N.FdetError("preprocessing runs on the GPU only (no CPU fallback): move the frames to cuda")
        if tuple(x.shape[-2:]) != tuple(self.input_shape[1:]):
            return hp.resize_bilinear_norm(x, tuple(self.input_shape[1:]))
        if x.dtype == torch.uint8:
            return hp.u8_to_f32_norm(x)
        return x.float() / 255.0

    def forward_frames(self, x: torch.Tensor) -> torch.Tensor:
        """`self(self.resize(x) / 255.0)` (models/PoolResnet.py:91-98) for frames -> the (N,5,S,S) maps.  uint8 frames at the
        model resolution go to the stem AS THEY ARE in inference: the division by 255 is fused into the stem's staging
        (fdet_stem_fwd_ps_u8; identical results, the fp32 image is never written).  Everything else is
        `_stack_forward(_preprocess(x))`."""
        if x.dim() == 3:
            x = x.unsqueeze(0)
        fused = (x.is_cuda and x.dtype == torch.uint8 and tuple(x.shape[-2:]) == tuple(self.input_shape[1:])
                 and not self.training and getattr(self.engine, "u8_frames_ok", lambda: False)())
        if fused:
            names, params = self.named_stack_params()
            fused = not (torch.is_grad_enabled() and any(p.requires_grad for p in params))
        if not fused:
            return self._stack_forward(self._preprocess(x))
        P = {n: p.detach() for n, p in zip(names, params)}
        return self.engine.forward(x, P, None, save=False, u8_frames=True)[0]

    # -------------------------------------------------------------- HIP-graph replay of the demo path
    def graphed_predict(self, example: torch.Tensor) -> "GraphedPredict":
        """The launch-bound `forward(frames, predict=1)` path (preprocess -> conv stack -> decode -> NMS: ~20 small
        launches for two frames) captured ONCE into a HIP graph; each call copies the new frames into the static
        input, replays the graph and reads the box counts.  Same kernels, same results as `forward(x, predict=1)`;
        thresholds and weights are those at capture time (re-capture after changing them)."""
        return GraphedPredict(self, example)

    def to_torchscript(self, file_path=None):
        """`torch.jit.script(model)` of the reference's export scripts (convert_checkpoint_to_scripted_model.py:51-54): the
        scriptable twin of this model (shared parameters, same state-dict names), optionally saved."""
        from ..torchscript import to_torchscript
        return to_torchscript(self, file_path)

    # -------------------------------------------------------------- reference surface
    def summary(self):
        if self.input_shape is None:
            raise Exception("Please set 'input_shape'")
        n = sum(p.numel() for p in self.parameters())
        print(f"{type(self).__name__}: {n:,} parameters, input {tuple(self.input_shape)}, "
              f"grid {self.num_of_patches}x{self.num_of_patches}")

    def non_max_suppression(self, x):
        if len(x.shape) == 4:
            rows, counts = self.reduce_bounding_boxes.forward_batch(x)
            return split_rows(rows, counts)
        return self.reduce_bounding_boxes(x)

    def single_non_max_suppression(self, x):
        return self.reduce_bounding_boxes(x)

    @torch.no_grad()
    def predict(self, x, probability_threshold=0.5, iou_threshold=0.5):
        self.reduce_bounding_boxes = ReduceBoundingBoxes(
            probability_threshold=probability_threshold, iou_threshold=iou_threshold,
            input_shape=self.input_shape, num_of_patches=self.num_of_patches)
        x = self._preprocess(x)
        image = x
        x = self(x)
        bbxs = self.non_max_suppression(x)
        return image, bbxs[0]


def split_rows(rows: torch.Tensor, counts: torch.Tensor):
    """(B,K,5) rows + (B,) counts -> the reference's tuple of B per-image (k_i,5) tensors (models/BaseModel.py:47-49; an image
    without boxes gives the reference's empty (0,5) CPU tensor, datasets/utils.py:170).  ONE gather of the valid prefixes and
    one `torch.split` instead of B Python slices: at 256 images the slicing alone kept the GPU idle for ~0.4 ms per batch."""
    import numpy as np
    B, K = rows.shape[0], rows.shape[1]
    cl = counts.tolist()                                   # the one host read-back of the call
    total = sum(cl)
    none = torch.empty(0).reshape(0, 5)
    if total == 0:
        return tuple(none for _ in range(B))
    c = np.asarray(cl, dtype=np.int64)
    first = np.concatenate(([0], np.cumsum(c)[:-1]))       # position of image i's first row in the packed tensor
    idx = np.repeat(np.arange(B, dtype=np.int64) * K - first, c) + np.arange(total, dtype=np.int64)
    packed = rows.reshape(B * K, rows.shape[2]).index_select(0, torch.from_numpy(idx).to(rows.device))
    return tuple(p if n else none for p, n in zip(torch.split(packed, cl), cl))


class GraphedPredict:
    """See BaseModel.graphed_predict."""

    def __init__(self, model: BaseModel, example: torch.Tensor):
        if not example.is_cuda:
            raise hp.N.FdetError("graphed_predict runs on the GPU only: move the example frames to cuda")
        if model.training:
            raise hp.N.FdetError("graphed_predict captures the eval path: call model.eval() first")
        self.model = model
        self.static_in = example.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):                                 # warm-up outside the capture (lazy packs, allocator)
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.rows, self.counts = self._run()
        # the captured launches hold raw pointers into the engine's pooled PS buffers: keep those buffers alive for as long
        # as the graph is (the engine drops its pool when the batch size changes)
        eng = getattr(model, "_engine", None)
        self._pool_refs = [t for lst in getattr(eng, "_ps_pool", {}).values() for t in lst] if eng is not None else []

    def _run(self):
        m = self.model
        y = m.forward_frames(self.static_in)
        return m.reduce_bounding_boxes.forward_batch(y)

    def __call__(self, frames: torch.Tensor = None):
        """-> tuple of per-image (K_i,5) box tensors [score,x,y,w,h]."""
        if frames is not None:
            self.static_in.copy_(frames, non_blocking=True)
        self.graph.replay()
        return split_rows(self.rows, self.counts)

    def first(self, frames: torch.Tensor = None):
        """Boxes of image 0 only, as `forward(x, predict=1)` returns them (models/PoolResnet.py:103-104)."""
        if frames is not None:
            self.static_in.copy_(frames, non_blocking=True)
        self.graph.replay()
        k = int(self.counts[0])
        return self.rows[0, :k] if k else torch.empty(0).reshape(0, 5)
