"""Tensor-level wrappers over the C-ABI (include/fdet.h).  Each function validates shapes on
the host (a kernel launched on mismatched shapes can fault the GPU), allocates the outputs
with torch, and enqueues the HIP kernels on torch's current stream.  GPU tensors only.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _native as N
from ._native import check, lib, ptr, stream

I32 = torch.int32
F32 = torch.float32


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != F32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------------------------
# detection math
# ------------------------------------------------------------------------------------------
def encode_targets(boxes: Sequence[torch.Tensor], img_size: Tuple[int, int], num_of_patches: int,
                   device=None) -> torch.Tensor:
    """Batched WIDERFaceDataset.convert_bbx_to_feature_map (dataset.py:32-64).
    `boxes`: list of (n_i,5) tensors [conf,x,y,w,h] -> (B,5,S,S) on the GPU."""
    B = len(boxes)
    device = device or torch.device("cuda", torch.cuda.current_device())
    counts = [int(b.shape[0]) if b.numel() else 0 for b in boxes]
    offs = torch.zeros(B + 1, dtype=I32)
    offs[1:] = torch.cumsum(torch.tensor(counts, dtype=torch.int64), 0).to(I32)
    rows = [b.reshape(-1, 5).to(F32).cpu() for b in boxes if b.numel()]
    flat = torch.cat(rows, 0) if rows else torch.zeros(1, 5)
    flat_d = flat.contiguous().to(device)
    offs_d = offs.to(device)
    S = int(num_of_patches)
    out = torch.empty(B, 5, S, S, dtype=F32, device=device)
    if B:
        check(lib().fdet_encode_targets(ptr(flat_d), ptr(offs_d, I32), B, S, float(img_size[0]), float(img_size[1]),
                                        ptr(out), stream()), "fdet_encode_targets")
    return out


def yolo_loss_fwd_bwd(pred: torch.Tensor, gt: torch.Tensor, want_grad: bool = True, grad_scale: float = 1.0):
    """(B,5,S,S) x2 -> (loss_per_image (B,), loss_sum (1,), grad (B,5,S,S) or None)."""
    pred, gt = _f32(pred), _f32(gt)
    if pred.dim() != 4 or pred.shape[1] != 5 or pred.shape[2] != pred.shape[3] or pred.shape != gt.shape:
        raise ValueError(f"yolo_loss: expected matching (B,5,S,S) maps, got {tuple(pred.shape)} / {tuple(gt.shape)}")
    B, _, S, _ = pred.shape
    lpi = torch.empty(B, dtype=F32, device=pred.device)
    lsum = torch.empty(1, dtype=F32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    check(lib().fdet_yolo_loss_fwd_bwd(ptr(pred), ptr(gt), B, S, ptr(lpi), ptr(lsum), ptr(grad), float(grad_scale),
                                       stream()), "fdet_yolo_loss_fwd_bwd")
    return lpi, lsum, grad


def decode(maps: torch.Tensor, prob_threshold: float, img_w: float, img_h: float):
    maps = _f32(maps)
    B, C, S, S2 = maps.shape
    if C != 5 or S != S2:
        raise ValueError(f"decode: expected (B,5,S,S), got {tuple(maps.shape)}")
    scores = torch.zeros(B, S * S, dtype=F32, device=maps.device)
    boxes = torch.zeros(B, S * S, 4, dtype=F32, device=maps.device)
    counts = torch.zeros(B, dtype=I32, device=maps.device)
    check(lib().fdet_decode(ptr(maps), B, S, float(prob_threshold), float(img_w), float(img_h), ptr(scores),
                            ptr(boxes), ptr(counts, I32), stream()), "fdet_decode")
    return scores, boxes, counts


def nms_batched(boxes: torch.Tensor, scores: torch.Tensor, counts: torch.Tensor, iou_threshold: float):
    boxes, scores = _f32(boxes), _f32(scores)
    B, K, four = boxes.shape
    if four != 4 or scores.shape != (B, K) or counts.shape != (B,):
        raise ValueError("nms_batched: expected boxes (B,K,4), scores (B,K), counts (B,)")
    counts = counts.to(I32).contiguous()
    keep = torch.zeros(B, K, dtype=I32, device=boxes.device)
    kc = torch.zeros(B, dtype=I32, device=boxes.device)
    check(lib().fdet_nms(ptr(boxes), ptr(scores), ptr(counts, I32), B, K, float(iou_threshold), ptr(keep, I32),
                         ptr(kc, I32), stream()), "fdet_nms")
    return keep, kc


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Drop-in for torchvision.ops.nms (keyword names as the reference passes them,
    datasets/utils.py:164): (K,4) xyxy, (K,) -> int64 kept indices by descending score."""
    K = boxes.shape[0]
    if K == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    if K > 4864:
        raise N.FdetError(f"nms: K={K} > 4864 candidates per image is not supported by fdet_nms")
    cnt = torch.tensor([K], dtype=I32, device=boxes.device)
    keep, kc = nms_batched(boxes.reshape(1, K, 4), scores.reshape(1, K), cnt, iou_threshold)
    return keep[0, : int(kc[0])].to(torch.int64)


def reduce_bounding_boxes(maps: torch.Tensor, prob_threshold: float, iou_threshold: float, img_w: float,
                          img_h: float):
    """Batched ReduceBoundingBoxes.forward: (B,5,S,S) -> (out (B,S*S,5), counts (B,))."""
    maps = _f32(maps)
    B, C, S, S2 = maps.shape
    if C != 5 or S != S2:
        raise ValueError(f"reduce_bounding_boxes: expected (B,5,S,S), got {tuple(maps.shape)}")
    out = torch.zeros(B, S * S, 5, dtype=F32, device=maps.device)
    counts = torch.zeros(B, dtype=I32, device=maps.device)
    check(lib().fdet_reduce_bounding_boxes(ptr(maps), B, S, float(prob_threshold), float(iou_threshold),
                                           float(img_w), float(img_h), ptr(out), ptr(counts, I32), stream()),
          "fdet_reduce_bounding_boxes")
    return out, counts


def step_metrics(gt: torch.Tensor, gt_counts: torch.Tensor, pred: torch.Tensor, pred_counts: torch.Tensor):
    B, K, five = gt.shape
    if five != 5 or pred.shape != gt.shape:
        raise ValueError("step_metrics: expected gt/pred (B,K,5)")
    per = torch.empty(B, 3, dtype=F32, device=gt.device)
    tot = torch.empty(3, dtype=F32, device=gt.device)
    check(lib().fdet_step_metrics(ptr(_f32(gt)), ptr(gt_counts.to(I32).contiguous(), I32), ptr(_f32(pred)),
                                  ptr(pred_counts.to(I32).contiguous(), I32), B, K, ptr(per), ptr(tot), stream()),
          "fdet_step_metrics")
    return per, tot


# ------------------------------------------------------------------------------------------
# SSD detection math (datasets/WIDERFace/dataset_ssd.py, losses/SSDLoss.py, datasets/utils.py:8-92)
# ------------------------------------------------------------------------------------------
SSD_PATCH_SIZES = (60, 30, 15, 7)


def _ps_array(patch_sizes):
    import ctypes
    ps = [int(p) for p in patch_sizes]
    return (ctypes.c_int * len(ps))(*ps), len(ps)


def ssd_num_priors(patch_sizes=SSD_PATCH_SIZES) -> int:
    arr, n = _ps_array(patch_sizes)
    P = int(lib().fdet_ssd_num_priors(arr, n))
    if P < 0:
        raise N.FdetError("ssd_num_priors: bad patch sizes")
    return P


def ssd_encode_targets(boxes: Sequence[torch.Tensor], img_size: Tuple[int, int], patch_sizes=SSD_PATCH_SIZES,
                       device=None) -> torch.Tensor:
    """Batched multi-scale SSD target encode: list of (n_i,5) [conf,x,y,w,h] -> (B,P,5) on the GPU."""
    B = len(boxes)
    device = device or torch.device("cuda", torch.cuda.current_device())
    counts = [int(b.shape[0]) if b.numel() else 0 for b in boxes]
    offs = torch.zeros(B + 1, dtype=I32)
    offs[1:] = torch.cumsum(torch.tensor(counts, dtype=torch.int64), 0).to(I32)
    rows = [b.reshape(-1, 5).to(F32).cpu() for b in boxes if b.numel()]
    flat = (torch.cat(rows, 0) if rows else torch.zeros(1, 5)).contiguous().to(device)
    arr, ns = _ps_array(patch_sizes)
    out = torch.empty(B, ssd_num_priors(patch_sizes), 5, dtype=F32, device=device)
    if B:
        check(lib().fdet_ssd_encode_targets(ptr(flat), ptr(offs.to(device), I32), B, arr, ns, float(img_size[0]),
                                            float(img_size[1]), ptr(out), stream()), "fdet_ssd_encode_targets")
    return out


def ssd_loss_fwd_bwd(pred: torch.Tensor, target: torch.Tensor, neg_pos_ratio: int = 10, want_grad: bool = True,
                     want_mask: bool = False):
    """ssd_loss(pred[:,:,0], pred[:,:,1:], target[:,:,0], target[:,:,1:], ratio): (loss (1,), grad or None, mask or None)."""
    pred, target = _f32(pred), _f32(target)
    if pred.dim() != 3 or pred.shape[2] != 5 or pred.shape != target.shape:
        raise ValueError(f"ssd_loss: expected matching (B,P,5) tensors, got {tuple(pred.shape)} / {tuple(target.shape)}")
    B, P, _ = pred.shape
    loss = torch.empty(1, dtype=F32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    mask = torch.empty(B, P, dtype=torch.uint8, device=pred.device) if want_mask else None
    ws = torch.empty(int(lib().fdet_ssd_loss_ws_bytes(B)) // 8 + 1, dtype=torch.float64, device=pred.device)
    check(lib().fdet_ssd_loss_fwd_bwd(ptr(pred), ptr(target), B, P, int(neg_pos_ratio), ptr(loss), ptr(grad),
                                      ptr(mask, torch.uint8), ptr(ws, torch.float64), ws.numel() * 8, stream()),
          "fdet_ssd_loss_fwd_bwd")
    return loss, grad, mask


def ssd_loss_parts(pred: torch.Tensor, target: torch.Tensor, neg_pos_ratio: int = 10, want_grad: bool = True):
    """Data-parallel half of ssd_loss: (sums (3,) f64 = [BCE, smooth-L1, positives] of THIS shard, UNSCALED grad or None)."""
    pred, target = _f32(pred), _f32(target)
    if pred.dim() != 3 or pred.shape[2] != 5 or pred.shape != target.shape:
        raise ValueError(f"ssd_loss: expected matching (B,P,5) tensors, got {tuple(pred.shape)} / {tuple(target.shape)}")
    B, P, _ = pred.shape
    sums = torch.empty(3, dtype=torch.float64, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    ws = torch.empty(int(lib().fdet_ssd_loss_ws_bytes(B)) // 8 + 1, dtype=torch.float64, device=pred.device)
    check(lib().fdet_ssd_loss_parts(ptr(pred), ptr(target), B, P, int(neg_pos_ratio), ptr(grad), None, ptr(sums, torch.float64),
                                    ptr(ws, torch.float64), ws.numel() * 8, stream()), "fdet_ssd_loss_parts")
    return sums, grad


def ssd_loss_finish(sums: torch.Tensor, grad: Optional[torch.Tensor]) -> torch.Tensor:
    """loss (1,) from the batch-wide sums (after the caller's SUM all-reduce); scales `grad` in place by 1 / positives."""
    if sums.dtype != torch.float64 or sums.numel() != 3:
        raise ValueError("ssd_loss_finish: sums must be the (3,) float64 tensor of ssd_loss_parts")
    loss = torch.empty(1, dtype=F32, device=sums.device)
    ws = torch.empty(4, dtype=F32, device=sums.device)
    check(lib().fdet_ssd_loss_finish(ptr(sums, torch.float64), ptr(loss), ptr(grad), grad.numel() if grad is not None else 0,
                                     ptr(ws), 16, stream()), "fdet_ssd_loss_finish")
    return loss


def ssd_reduce_bounding_boxes(x: torch.Tensor, prob_threshold: float, iou_threshold: float, img_w: float, img_h: float,
                              patch_sizes=SSD_PATCH_SIZES, with_priors: bool = True, priors: torch.Tensor = None):
    """Batched ReduceSSDBoundingBoxes.forward: (B,P,5) -> (rows (B,P,5), counts (B,)).  priors: optional (P,4) table
    (datasets/utils.py:31-32), None = the reference's calculate_priors()."""
    x = _f32(x)
    B, P, five = x.shape
    if five != 5 or P != ssd_num_priors(patch_sizes):
        raise ValueError(f"ssd_reduce_bounding_boxes: expected (B,{ssd_num_priors(patch_sizes)},5), got {tuple(x.shape)}")
    arr, ns = _ps_array(patch_sizes)
    out = torch.zeros(B, P, 5, dtype=F32, device=x.device)
    counts = torch.zeros(B, dtype=I32, device=x.device)
    if priors is not None:
        priors = _f32(priors.to(x.device))
        if tuple(priors.shape) != (P, 4):
            raise ValueError(f"ssd_reduce_bounding_boxes: priors must be ({P},4), got {tuple(priors.shape)}")
    check(lib().fdet_ssd_reduce_bounding_boxes_priors(ptr(x), B, arr, ns, int(bool(with_priors)), ptr(priors), float(prob_threshold),
                                                      float(iou_threshold), float(img_w), float(img_h), ptr(out),
                                                      ptr(counts, I32), stream()), "fdet_ssd_reduce_bounding_boxes")
    return out, counts


def ssd_head_pack_fwd(z: torch.Tensor, ps: int, prior_start: int, y: torch.Tensor) -> None:
    Nn, CP, H, W = z.shape
    if H != ps or W != ps or y.dim() != 3 or y.shape[0] != Nn or y.shape[2] != 5:
        raise ValueError("ssd_head_pack_fwd: shape mismatch")
    check(lib().fdet_ssd_head_pack_fwd(ptr(z), Nn, CP, ps, int(prior_start), y.shape[1], ptr(y), stream()), "fdet_ssd_head_pack_fwd")


def ssd_head_pack_bwd(dy: torch.Tensor, y: torch.Tensor, ps: int, prior_start: int, dz: torch.Tensor) -> None:
    Nn, CP, H, W = dz.shape
    if H != ps or W != ps or dy.shape != y.shape or y.shape[0] != Nn or y.shape[2] != 5:
        raise ValueError("ssd_head_pack_bwd: shape mismatch")
    check(lib().fdet_ssd_head_pack_bwd(ptr(dy), ptr(y), Nn, CP, ps, int(prior_start), y.shape[1], ptr(dz), stream()), "fdet_ssd_head_pack_bwd")


def u8_to_f32_norm(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.dtype != torch.uint8:
        raise TypeError("u8_to_f32_norm expects uint8")
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=F32, device=x.device)
    elif tuple(out.shape) != tuple(x.shape):
        raise ValueError("u8_to_f32_norm: out shape differs")
    check(lib().fdet_u8_to_f32_norm(ptr(x, torch.uint8), ptr(out), x.numel(), stream()), "fdet_u8_to_f32_norm")
    return out


def resize_bilinear_norm(x: torch.Tensor, size: Tuple[int, int], divisor: float = 255.0,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`Resize(size)(x) / 255.0` on the device (torchvision 0.11.2 tensor semantics: bilinear,
    align_corners=False, no antialias).  uint8 input goes through the uint8 round trip (round half
    even) before the division; float input is interpolated and divided.  (N,C,H,W) or (C,H,W)."""
    if x.dim() == 3:
        x = x.unsqueeze(0)
    if x.dim() != 4:
        raise ValueError(f"resize_bilinear_norm: expected (N,C,H,W) or (C,H,W), got {tuple(x.shape)}")
    x = x.contiguous()
    Nn, C, Hs, Ws = x.shape
    Hd, Wd = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty(Nn, C, Hd, Wd, dtype=F32, device=x.device)
    else:
        _chk4(out, (Nn, C, Hd, Wd), "out")
    if x.dtype == torch.uint8:
        if divisor != 255.0:
            raise ValueError("resize_bilinear_norm: the uint8 path always divides by 255")
        check(lib().fdet_resize_bilinear_u8_norm(ptr(x, torch.uint8), ptr(out), Nn, C, Hs, Ws, Hd, Wd, stream()),
              "fdet_resize_bilinear_u8_norm")
    else:
        check(lib().fdet_resize_bilinear_f32_norm(ptr(_f32(x)), ptr(out), Nn, C, Hs, Ws, Hd, Wd, float(divisor), stream()),
              "fdet_resize_bilinear_f32_norm")
    return out


# ------------------------------------------------------------------------------------------
# optimiser / dropout
# ------------------------------------------------------------------------------------------
def adam_step(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
              lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
              grad_scale: float = 1.0) -> None:
    n = param.numel()
    if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n):
        raise ValueError("adam_step: buffer sizes differ")
    check(lib().fdet_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), n, int(step), float(lr),
                               float(beta1), float(beta2), float(eps), float(grad_scale), stream()), "fdet_adam_step")


def dropout_scales(out: torch.Tensor, p: float, seed: int, offset: int) -> torch.Tensor:
    check(lib().fdet_dropout_scales(ptr(out), out.numel(), float(p), int(seed) & (2**64 - 1),
                                    int(offset) & (2**64 - 1), stream()), "fdet_dropout_scales")
    return out


def dropout_scales_layers(n: int, channels: Sequence[int], p: Sequence[float], seed: int, base: int, first_image: int,
                          device) -> List[torch.Tensor]:
    """Dropout2d scales of every dropout layer in one launch (fdet_dropout_scales_layers): returns one dense
    (n, channels[k]) tensor per layer (views into one buffer).  The counter is indexed by the global image number
    `first_image + i`, so data-parallel ranks draw what a single process draws for the concatenated batch."""
    import ctypes
    L = len(channels)
    if L != len(p) or not (1 <= L <= 32):
        raise ValueError("dropout_scales_layers: 1..32 layers, one p per layer")
    ls = int(sum(channels))
    buf = torch.empty(max(n * ls, 1), dtype=F32, device=device)
    ch = (ctypes.c_int * L)(*[int(c) for c in channels])
    pp = (ctypes.c_float * L)(*[float(v) for v in p])
    m64 = 2**64 - 1
    check(lib().fdet_dropout_scales_layers(ptr(buf), int(n), L, ch, pp, int(seed) & m64, int(base) & m64,
                                           int(first_image) & m64, stream()), "fdet_dropout_scales_layers")
    out, off = [], 0
    for c in channels:
        out.append(buf[off:off + n * c].view(n, c))
        off += n * c
    return out


# ------------------------------------------------------------------------------------------
# conv stack primitives (shapes are validated here; the kernels trust them)
# ------------------------------------------------------------------------------------------
def _chk4(t: torch.Tensor, shape, name):
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


def packed_sizes(cout: int, cin: int) -> Tuple[int, int]:
    cop, cip = (cout + 31) // 32 * 32, (cin + 31) // 32 * 32
    return cin * 9 * cop, cout * 9 * cip


def x3_supported(cout: int, cin: int) -> bool:
    """bf16x3 conv kernels need channel counts that are multiples of 16."""
    return cout % 16 == 0 and cin % 16 == 0


def pack_conv3x3_weights(w: torch.Tensor, wpk_fwd: Optional[torch.Tensor], wpk_bwd: Optional[torch.Tensor],
                         x3: bool = False) -> None:
    """x3=True packs the bf16 hi/lo panels of the bf16x3 kernels (same buffer sizes)."""
    cout, cin, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise ValueError("pack_conv3x3_weights: 3x3 kernels only")
    nf, nb = packed_sizes(cout, cin)
    if wpk_fwd is not None and wpk_fwd.numel() != nf:
        raise ValueError(f"wpk_fwd must hold {nf} floats")
    if wpk_bwd is not None and wpk_bwd.numel() != nb:
        raise ValueError(f"wpk_bwd must hold {nb} floats")
    fn = lib().fdet_pack_conv3x3_weights_bf16x3 if x3 else lib().fdet_pack_conv3x3_weights
    check(fn(ptr(w), cout, cin, ptr(wpk_fwd), ptr(wpk_bwd), stream()), "fdet_pack_conv3x3_weights")


def pack_conv3x3_weights_batched(ws, wpk_fwds, wpk_bwds) -> None:
    """bf16x3 panels of len(ws) same-shape layers in one launch."""
    cout, cin, kh, kw = ws[0].shape
    if (kh, kw) != (3, 3):
        raise ValueError("pack_conv3x3_weights_batched: 3x3 kernels only")
    nf, nb = packed_sizes(cout, cin)
    for w, f_, b_ in zip(ws, wpk_fwds, wpk_bwds):
        _chk4(w, (cout, cin, 3, 3), "w")
        if f_.numel() != nf or b_.numel() != nb:
            raise ValueError("pack_conv3x3_weights_batched: packed buffer size does not match (Cout,Cin)")
    import ctypes
    arr = ctypes.c_void_p * len(ws)
    check(lib().fdet_pack_conv3x3_weights_bf16x3_batched(arr(*[ptr(w) for w in ws]), len(ws), cout, cin,
                                                         arr(*[ptr(t) for t in wpk_fwds]), arr(*[ptr(t) for t in wpk_bwds]),
                                                         stream()), "fdet_pack_conv3x3_weights_bf16x3_batched")


def conv3x3_fwd(x, wpk, bias, cout: int, y_full=None, skip=None, drop_scale=None, y_out=None, slope: float = 0.2,
                x3: bool = False):
    Nn, cin, H, W = x.shape
    if wpk.numel() != packed_sizes(cout, cin)[0]:
        raise ValueError("conv3x3_fwd: packed weight size does not match (Cout,Cin)")
    for t, nm in ((y_full, "y_full"), (skip, "skip"), (y_out, "y_out")):
        if t is not None:
            _chk4(t, (Nn, cout, H, W), nm)
    if bias is not None:
        _chk4(bias, (cout,), "bias")
    if drop_scale is not None:
        _chk4(drop_scale, (Nn, cout), "drop_scale")
    fn = lib().fdet_conv3x3_fwd_bf16x3 if x3 else lib().fdet_conv3x3_fwd
    check(fn(ptr(x), ptr(wpk), ptr(bias), ptr(y_full), ptr(skip), ptr(drop_scale), ptr(y_out),
             Nn, cin, cout, H, W, 1, float(slope), stream()), "fdet_conv3x3_fwd")


def conv3x3_dgrad(dz, wpk_bwd, cin: int, dx, act=None, add=None, slope: float = 0.2, x3: bool = False):
    Nn, cout, H, W = dz.shape
    if wpk_bwd.numel() != packed_sizes(cout, cin)[1]:
        raise ValueError("conv3x3_dgrad: packed weight size does not match (Cout,Cin)")
    _chk4(dx, (Nn, cin, H, W), "dx")
    for t, nm in ((act, "act"), (add, "add")):
        if t is not None:
            _chk4(t, (Nn, cin, H, W), nm)
    fn = lib().fdet_conv3x3_dgrad_bf16x3 if x3 else lib().fdet_conv3x3_dgrad
    check(fn(ptr(dz), ptr(wpk_bwd), ptr(act), ptr(add), ptr(dx), Nn, cin, cout, H, W, float(slope), stream()),
          "fdet_conv3x3_dgrad")


def pool_fusion_supported(cout: int, cin: int, H: int, W: int, N: int = 1) -> bool:
    """Pooled residual blocks whose tail runs inside the conv epilogues (fdet_conv3x3_fwd_pool_bf16x3): the library's own
    plan check (fdet_conv3x3_pool_fusion_ok), batch-size limits included when N is given."""
    return bool(lib().fdet_conv3x3_pool_fusion_ok(int(N), int(cin), int(cout), int(H), int(W)))


def conv3x3_fwd_pool(x, wpk, bias, skip, drop_scale, out_pooled, route, slope: float = 0.2):
    """out_pooled = maxpool2x2(lrelu(conv(x)+bias)*drop_scale + skip); route: uint8 routing bytes or None (eval)."""
    Nn, cin, H, W = x.shape
    cout = bias.shape[0]
    if wpk.numel() != packed_sizes(cout, cin)[0]:
        raise ValueError("conv3x3_fwd_pool: packed weight size does not match (Cout,Cin)")
    _chk4(skip, (Nn, cout, H, W), "skip")
    _chk4(out_pooled, (Nn, cout, H // 2, W // 2), "out_pooled")
    if route is not None:
        _chk4(route, (Nn, cout, H // 2, W // 2), "route")
    if drop_scale is not None:
        _chk4(drop_scale, (Nn, cout), "drop_scale")
    check(lib().fdet_conv3x3_fwd_pool_bf16x3(ptr(x), ptr(wpk), ptr(bias), ptr(skip), ptr(drop_scale), ptr(out_pooled),
                                             ptr(route, torch.uint8), Nn, cin, cout, H, W, float(slope), stream()),
          "fdet_conv3x3_fwd_pool_bf16x3")


def conv3x3_dgrad_unpool(dz, wpk_bwd, cin: int, dout_pooled, route, dx, slope: float = 0.2):
    """dx = conv^T(dz) + unpool(dout_pooled) through the routing bytes of the forward pass."""
    Nn, cout, H, W = dz.shape
    if wpk_bwd.numel() != packed_sizes(cout, cin)[1]:
        raise ValueError("conv3x3_dgrad_unpool: packed weight size does not match (Cout,Cin)")
    _chk4(dx, (Nn, cin, H, W), "dx")
    _chk4(dout_pooled, (Nn, cin, H // 2, W // 2), "dout_pooled")
    _chk4(route, (Nn, cin, H // 2, W // 2), "route")
    check(lib().fdet_conv3x3_dgrad_unpool_bf16x3(ptr(dz), ptr(wpk_bwd), ptr(dout_pooled), ptr(route, torch.uint8), ptr(dx),
                                                 Nn, cin, cout, H, W, float(slope), stream()),
          "fdet_conv3x3_dgrad_unpool_bf16x3")


def pool_route_bwd(dout_pooled, route, drop_scale, dz2, slope: float = 0.2):
    """dz2 = unpool(dout_pooled) * drop_scale * lrelu'(c), from the routing bytes alone."""
    Nn, F_, H, W = dz2.shape
    _chk4(dout_pooled, (Nn, F_, H // 2, W // 2), "dout_pooled")
    _chk4(route, (Nn, F_, H // 2, W // 2), "route")
    if drop_scale is not None:
        _chk4(drop_scale, (Nn, F_), "drop_scale")
    check(lib().fdet_pool_route_bwd(ptr(dout_pooled), ptr(route, torch.uint8), ptr(drop_scale), ptr(dz2), Nn, F_, H, W,
                                    float(slope), stream()), "fdet_pool_route_bwd")


# ---- pointwise (1x1 conv / per-position Linear) GEMMs, bf16x3 ------------------------------------------------------
def pointwise_pack(w: torch.Tensor):
    """w (Cout,Cin[,1,1]) -> (forward panel, backward panel) for pointwise_fwd / pointwise_dgrad."""
    cout, cin = int(w.shape[0]), int(w.shape[1])
    w2 = _f32(w.reshape(cout, cin))
    n = int(lib().fdet_pointwise_packed_bytes(cout, cin)) // 4
    f_ = torch.empty(n, dtype=F32, device=w.device)
    b_ = torch.empty(n, dtype=F32, device=w.device)
    check(lib().fdet_pack_pointwise_weights_bf16x3(ptr(w2), cout, cin, ptr(f_), ptr(b_), stream()), "fdet_pack_pointwise_weights_bf16x3")
    return f_, b_


def pointwise_fwd(x, wpk_fwd, bias, y, slope: float = 1.0):
    """y (N,Cout,H,W) = lrelu_slope(W x + bias) for x (N,Cin,H,W)."""
    Nn, cin = x.shape[0], x.shape[1]
    cout = y.shape[1]
    P = x.numel() // (Nn * cin)
    if y.shape[0] != Nn or y.numel() != Nn * cout * P:
        raise ValueError("pointwise_fwd: x / y shapes disagree")
    if bias is not None:
        _chk4(bias, (cout,), "bias")
    if wpk_fwd.numel() * 4 < int(lib().fdet_pointwise_packed_bytes(cout, cin)):
        raise ValueError("pointwise_fwd: packed weight buffer too small for (Cout,Cin)")
    check(lib().fdet_pointwise_fwd_bf16x3(ptr(x), ptr(wpk_fwd), ptr(bias), ptr(y), Nn, cin, cout, P, float(slope), stream()),
          "fdet_pointwise_fwd_bf16x3")


def pointwise_dgrad(dz, wpk_bwd, dx, add=None):
    """dx (N,Cin,H,W) = W^T dz (+ add)."""
    Nn, cout = dz.shape[0], dz.shape[1]
    cin = dx.shape[1]
    P = dz.numel() // (Nn * cout)
    if dx.shape[0] != Nn or dx.numel() != Nn * cin * P or (add is not None and tuple(add.shape) != tuple(dx.shape)):
        raise ValueError("pointwise_dgrad: shapes disagree")
    if wpk_bwd.numel() * 4 < int(lib().fdet_pointwise_packed_bytes(cout, cin)):
        raise ValueError("pointwise_dgrad: packed weight buffer too small for (Cout,Cin)")
    check(lib().fdet_pointwise_dgrad_bf16x3(ptr(dz), ptr(wpk_bwd), ptr(add), ptr(dx), Nn, cin, cout, P, stream()),
          "fdet_pointwise_dgrad_bf16x3")


def pointwise_wgrad(x, dz, dW, db=None):
    """dW (Cout,Cin[,1,1]) = sum dz x^T, db (Cout,) = sum dz."""
    Nn, cin = x.shape[0], x.shape[1]
    cout = dz.shape[1]
    P = x.numel() // (Nn * cin)
    if dz.shape[0] != Nn or dz.numel() != Nn * cout * P or dW.numel() != cout * cin or (db is not None and db.numel() != cout):
        raise ValueError("pointwise_wgrad: shapes disagree")
    nb = int(lib().fdet_pointwise_wgrad_ws_bytes(Nn, cin, cout, P))
    ws = torch.empty(nb // 4 + 4, dtype=F32, device=x.device)
    check(lib().fdet_pointwise_wgrad_bf16x3(ptr(x), ptr(dz), ptr(dW), ptr(db), ptr(ws), ws.numel() * 4, Nn, cin, cout, P, stream()),
          "fdet_pointwise_wgrad_bf16x3")


def conv3x3_wgrad_ws_bytes(Nn, cin, cout, H, W) -> int:
    """Workspace that serves BOTH precisions of conv3x3_wgrad."""
    return max(int(lib().fdet_conv3x3_wgrad_ws_bytes(Nn, cin, cout, H, W)),
               int(lib().fdet_conv3x3_wgrad_bf16x3_ws_bytes(Nn, cin, cout, H, W)))


def wgrad_x3_supported(Nn, cin, cout, H, W) -> bool:
    return int(lib().fdet_conv3x3_wgrad_bf16x3_ws_bytes(Nn, cin, cout, H, W)) > 0


def conv3x3_wgrad(x, dz, dW, db, ws, x3: bool = False):
    Nn, cin, H, W = x.shape
    cout = dz.shape[1]
    _chk4(dz, (Nn, cout, H, W), "dz")
    _chk4(dW, (cout, cin, 3, 3), "dW")
    _chk4(db, (cout,), "db")
    fn = lib().fdet_conv3x3_wgrad_bf16x3 if x3 else lib().fdet_conv3x3_wgrad
    check(fn(ptr(x), ptr(dz), ptr(dW), ptr(db), ptr(ws, ws.dtype), ws.numel() * ws.element_size(),
             Nn, cin, cout, H, W, stream()), "fdet_conv3x3_wgrad")


def conv3x3_wgrad_batched_ws_bytes(L, Nn, cin, cout, H, W) -> int:
    return int(lib().fdet_conv3x3_wgrad_bf16x3_batched_ws_bytes(L, Nn, cin, cout, H, W))


def conv3x3_wgrad_batched(xs, dzs, dWs, dbs, ws):
    """bf16x3 weight gradients of L <= 16 same-shape layers in one launch (+ one reduce)."""
    L = len(xs)
    if not (1 <= L <= 16 and len(dzs) == len(dWs) == len(dbs) == L):
        raise ValueError("conv3x3_wgrad_batched: 1..16 layers, equal list lengths")
    Nn, cin, H, W = xs[0].shape
    cout = dzs[0].shape[1]
    for x, dz, dW, db in zip(xs, dzs, dWs, dbs):
        _chk4(x, (Nn, cin, H, W), "x")
        _chk4(dz, (Nn, cout, H, W), "dz")
        _chk4(dW, (cout, cin, 3, 3), "dW")
        _chk4(db, (cout,), "db")
    import ctypes
    arr = ctypes.c_void_p * L
    hx, hdz = arr(*[ptr(t) for t in xs]), arr(*[ptr(t) for t in dzs])
    hdW, hdb = arr(*[ptr(t) for t in dWs]), arr(*[ptr(t) for t in dbs])
    check(lib().fdet_conv3x3_wgrad_bf16x3_batched(hx, hdz, hdW, hdb, L, ptr(ws, ws.dtype), ws.numel() * ws.element_size(),
                                                  Nn, cin, cout, H, W, stream()), "fdet_conv3x3_wgrad_bf16x3_batched")


def _ptr_array(ts, dtype=F32):
    """HOST array of device pointers (None entries -> NULL); None -> NULL array."""
    import ctypes
    if ts is None:
        return None
    return (ctypes.c_void_p * len(ts))(*[ptr(t, dtype) for t in ts])


def block_chain_supported(F_: int, H: int, W: int) -> bool:
    return bool(lib().fdet_block_chain_supported(int(F_), int(H), int(W)))


def block_chain_fwd(x, wpk1, b1, wpk2, b2, scales, a_out, c_out, outs, slope: float = 0.2):
    """Run len(wpk1) un-pooled residual blocks in one launch (fdet_block_chain_fwd_bf16x3).
    Lists of per-block tensors; `scales`, `a_out`, `c_out` may be None; `outs` entries may be None
    except the last."""
    nb = len(wpk1)
    Nn, F_, H, W = x.shape
    if not (len(b1) == len(wpk2) == len(b2) == len(outs) == nb) or outs[-1] is None:
        raise ValueError("block_chain_fwd: inconsistent per-block lists")
    for lst, nm in ((a_out, "a_out"), (c_out, "c_out"), (outs, "outs")):
        for t in (lst or []):
            if t is not None:
                _chk4(t, (Nn, F_, H, W), nm)
    for t in (scales or []):
        if t is not None:
            _chk4(t, (Nn, F_), "scale")
    nf = packed_sizes(F_, F_)[0]
    for t in list(wpk1) + list(wpk2):
        if t.numel() != nf:
            raise ValueError("block_chain_fwd: packed weight size does not match the channel count")
    for t in list(b1) + list(b2):
        _chk4(t, (F_,), "bias")
    check(lib().fdet_block_chain_fwd_bf16x3(ptr(x), _ptr_array(wpk1), _ptr_array(b1), _ptr_array(wpk2), _ptr_array(b2),
                                            _ptr_array(scales), _ptr_array(a_out), _ptr_array(c_out), _ptr_array(outs),
                                            nb, Nn, F_, H, W, float(slope), stream()), "fdet_block_chain_fwd_bf16x3")


def block_chain_bwd(dout, wpk1b, wpk2b, scales, a_saved, c_saved, dz1, dz2, dx, slope: float = 0.2):
    """Data-gradient chain of the same blocks (fdet_block_chain_bwd_bf16x3): fills dz1[k], dz2[k], dx."""
    nb = len(wpk1b)
    Nn, F_, H, W = dout.shape
    if not (len(wpk2b) == len(a_saved) == len(c_saved) == len(dz1) == len(dz2) == nb):
        raise ValueError("block_chain_bwd: inconsistent per-block lists")
    for lst, nm in ((a_saved, "a"), (c_saved, "c"), (dz1, "dz1"), (dz2, "dz2"), ([dx], "dx")):
        for t in lst:
            _chk4(t, (Nn, F_, H, W), nm)
    for t in (scales or []):
        if t is not None:
            _chk4(t, (Nn, F_), "scale")
    nbk = packed_sizes(F_, F_)[1]
    for t in list(wpk1b) + list(wpk2b):
        if t.numel() != nbk:
            raise ValueError("block_chain_bwd: packed weight size does not match the channel count")
    check(lib().fdet_block_chain_bwd_bf16x3(ptr(dout), _ptr_array(wpk1b), _ptr_array(wpk2b), _ptr_array(scales),
                                            _ptr_array(a_saved), _ptr_array(c_saved), _ptr_array(dz1), _ptr_array(dz2),
                                            ptr(dx), nb, Nn, F_, H, W, float(slope), stream()), "fdet_block_chain_bwd_bf16x3")


def block_tail_fwd(c, x, drop_scale, out, pool: int):
    Nn, F_, H, W = c.shape
    _chk4(x, c.shape, "x")
    _chk4(out, (Nn, F_, H // pool, W // pool), "out")
    if drop_scale is not None:
        _chk4(drop_scale, (Nn, F_), "drop_scale")
    check(lib().fdet_block_tail_fwd(ptr(c), ptr(x), ptr(drop_scale), ptr(out), Nn, F_, H, W, pool, stream()),
          "fdet_block_tail_fwd")


def block_tail_bwd(dout, c, x, drop_scale, dz2, de, pool: int, slope: float = 0.2):
    Nn, F_, H, W = c.shape
    _chk4(dout, (Nn, F_, H // pool, W // pool), "dout")
    _chk4(dz2, c.shape, "dz2")
    if pool == 2:
        _chk4(x, c.shape, "x")
        _chk4(de, c.shape, "de")
    if drop_scale is not None:
        _chk4(drop_scale, (Nn, F_), "drop_scale")
    check(lib().fdet_block_tail_bwd(ptr(dout), ptr(c), ptr(x), ptr(drop_scale), ptr(dz2), ptr(de), Nn, F_, H, W, pool,
                                    float(slope), stream()), "fdet_block_tail_bwd")


def stem_ws_bytes(Nn, cin, F_, H, W, k, stride, pad) -> int:
    return int(lib().fdet_stem_ws_bytes(Nn, cin, F_, H, W, k, stride, pad))


def stem_x3_supported(cin, W, k, stride, pad) -> bool:
    return cin == 3 and (k, stride, pad) == (10, 8, 2) and W % 4 == 0 and W <= 512 and ((W + 2 * pad - k) // stride + 1) % 4 == 0


def stem_fwd(x, w, bias, y, ws, k, stride, pad, x3: bool = False):
    Nn, cin, H, W = x.shape
    F_ = w.shape[0]
    _chk4(w, (F_, cin, k, k), "w")
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    _chk4(y, (Nn, F_, Ho, Wo), "y")
    fn = lib().fdet_stem_fwd_bf16x3 if x3 else lib().fdet_stem_fwd
    check(fn(ptr(x), ptr(w), ptr(bias), ptr(y), ptr(ws, ws.dtype), ws.numel() * ws.element_size(),
             Nn, cin, F_, H, W, k, stride, pad, stream()), "fdet_stem_fwd")


def stem_k3_wgrad_x3_supported(cin, F_, H, W, k, stride, pad) -> bool:
    """The Resnet stem (3 channels, k3 s2 p1) has a matrix-core (bf16x3) weight gradient for this shape (fdet_stem_k3.hip)."""
    Wo = (W + 2 * pad - k) // stride + 1
    return cin == 3 and (k, stride, pad) == (3, 2, 1) and F_ % 8 == 0 and H % 2 == 0 and W % 4 == 0 and Wo % 16 == 0 and Wo <= 320


def stem_k3_fwd_ps_supported(cin, F_, H, W, k, stride, pad) -> bool:
    return cin == 3 and (k, stride, pad) == (3, 2, 1) and F_ % 8 == 0 and H % 2 == 0 and W % 2 == 0


def stem_wgrad(x, dy, dW, db, ws, k, stride, pad, x3: bool = False, p16: bool = False):
    Nn, cin, H, W = x.shape
    F_ = dW.shape[0]
    _chk4(dW, (F_, cin, k, k), "dW")
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    _chk4(dy, (Nn, F_, Ho, Wo), "dy")
    fn = lib().fdet_stem_wgrad_bf16 if (x3 and p16) else (lib().fdet_stem_wgrad_bf16x3 if x3 else lib().fdet_stem_wgrad)
    check(fn(ptr(x), ptr(dy), ptr(dW), ptr(db), ptr(ws, ws.dtype), ws.numel() * ws.element_size(),
             Nn, cin, F_, H, W, k, stride, pad, stream()), "fdet_stem_wgrad")


def head_fwd(x, drop_scale, w, bias, y, k, pad):
    Nn, F_, H, W = x.shape
    _chk4(w, (5, F_, k, k), "w")
    So, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    _chk4(y, (Nn, 5, So, Wo), "y")
    if drop_scale is not None:
        _chk4(drop_scale, (Nn, F_), "drop_scale")
    check(lib().fdet_head_fwd(ptr(x), ptr(drop_scale), ptr(w), ptr(bias), ptr(y), Nn, F_, H, W, k, pad, stream()),
          "fdet_head_fwd")


def head_bwd_ws_bytes(Nn, F_, H, W, k, pad) -> int:
    return int(lib().fdet_head_bwd_ws_bytes(Nn, F_, H, W, k, pad))


def head_bwd(x, drop_scale, w, y, dy, dx, dW, db, ws, k, pad):
    Nn, F_, H, W = x.shape
    _chk4(w, (5, F_, k, k), "w")
    _chk4(dW, (5, F_, k, k), "dW")
    _chk4(dx, x.shape, "dx")
    _chk4(dy, y.shape, "dy")
    check(lib().fdet_head_bwd(ptr(x), ptr(drop_scale), ptr(w), ptr(y), ptr(dy), ptr(dx), ptr(dW), ptr(db),
                              ptr(ws, ws.dtype), ws.numel() * ws.element_size(), Nn, F_, H, W, k, pad, stream()),
          "fdet_head_bwd")


def head_loss_fused_supported(F_, H, W, k, pad) -> bool:
    return bool(lib().fdet_head_loss_fused_supported(int(F_), int(H), int(W), int(k), int(pad)))


def head_loss_fused_ws_bytes(Nn, F_, H, W, k, pad) -> int:
    return int(lib().fdet_head_loss_fused_ws_bytes(Nn, F_, H, W, k, pad))


def head_loss_fused(x, drop_scale, w, bias, gt, y, lpi, lsum, dx, dW, db, ws, k, pad):
    """Training head + yolo_loss + their gradients in one launch sequence (fdet_head_loss_fused): fills y (N,5,S,S),
    lpi (N,), lsum (1,), dx like x, dW (5,F,k,k), db (5,).  ws: zero-initialised workspace of head_loss_fused_ws_bytes()."""
    Nn, F_, H, W = x.shape
    _chk4(w, (5, F_, k, k), "w")
    _chk4(dW, (5, F_, k, k), "dW")
    _chk4(dx, x.shape, "dx")
    _chk4(gt, y.shape, "gt")
    if drop_scale is not None:
        _chk4(drop_scale, (Nn, F_), "drop_scale")
    check(lib().fdet_head_loss_fused(ptr(x), ptr(drop_scale), ptr(w), ptr(bias), ptr(gt), ptr(y), ptr(lpi), ptr(lsum), ptr(dx),
                                     ptr(dW), ptr(db), ptr(ws, ws.dtype), ws.numel() * ws.element_size(), Nn, F_, H, W, k, pad,
                                     stream()), "fdet_head_loss_fused")


# ---- MobileNetV3 backbone (inference, bf16, NHWC activations) -------------------------------------------------------
BF16 = torch.bfloat16
MB_ACT = {"none": 0, "relu": 1, "hswish": 2}


def mb_stem(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """x (N,3,H,W) f32 in [0,1] or uint8 (the /255 fused) -> (N,H/2,W/2,16) bf16; w (16,27) f32 BN-folded."""
    if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 2 or x.shape[3] % 2 or x.dtype not in (F32, torch.uint8):
        raise ValueError(f"mb_stem: expected (N,3,H,W) f32/uint8 with even H, W; got {tuple(x.shape)} {x.dtype}")
    _chk4(w, (16, 27), "stem weight"); _chk4(bias, (16,), "stem bias")
    Nn, _, H, W = x.shape
    y = torch.empty(Nn, H // 2, W // 2, 16, dtype=BF16, device=x.device)
    check(lib().fdet_mb_stem(ptr(x, x.dtype), int(x.dtype == torch.uint8), ptr(w), ptr(bias), ptr(y, BF16), Nn, H, W, stream()),
          "fdet_mb_stem")
    return y


def mb_depthwise(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, K: int, stride: int, act: int, want_pool: bool):
    """x (N,H,W,C) bf16 -> (y (N,Ho,Wo,C) bf16, per-image partial channel sums (N,slots,C) f32 or None); w (K*K,C) f32."""
    if x.dim() != 4 or x.dtype != BF16:
        raise ValueError("mb_depthwise: expected (N,H,W,C) bf16")
    Nn, H, W, C = x.shape
    _chk4(w, (K * K, C), "depthwise weight"); _chk4(bias, (C,), "depthwise bias")
    Ho, Wo = -(-H // stride), -(-W // stride)
    y = torch.empty(Nn, Ho, Wo, C, dtype=BF16, device=x.device)
    pool = None
    if want_pool:
        slots = int(lib().fdet_mb_depthwise_pool_slots(Nn, H, W, C, K, stride))
        if slots <= 0:
            raise ValueError("mb_depthwise: unsupported shape")
        pool = torch.empty(Nn, slots, C, dtype=F32, device=x.device)
    check(lib().fdet_mb_depthwise(ptr(x, BF16), ptr(w), ptr(bias), ptr(y, BF16), ptr(pool), Nn, H, W, C, K, stride, act, stream()),
          "fdet_mb_depthwise")
    return y, pool


def mb_se_gate(pool: torch.Tensor, HW: int, w1, b1, w2, b2) -> torch.Tensor:
    """gate (N,C) = hardsigmoid(W2 relu(W1 mean + b1) + b2), mean = sum of the pool rows / HW; pool (N,slots,C) as
    mb_depthwise returns it (or (N,C)); w1 (R,C), w2 (C,R)."""
    if pool.dim() == 2:
        pool = pool.unsqueeze(1)
    Nn, slots, C = pool.shape
    R = w1.shape[0]
    _chk4(w1, (R, C), "se reduce weight"); _chk4(b1, (R,), "se reduce bias")
    _chk4(w2, (C, R), "se expand weight"); _chk4(b2, (C,), "se expand bias")
    gate = torch.empty(Nn, C, dtype=F32, device=pool.device)
    check(lib().fdet_mb_se_gate(ptr(pool), slots, HW, ptr(w1), ptr(b1), ptr(w2), ptr(b2), Nn, C, R, ptr(gate), stream()), "fdet_mb_se_gate")
    return gate


def mb_pointwise_pack(w: torch.Tensor, bias: torch.Tensor):
    """BN-folded (Cout,Cin) f32 weight + (Cout,) bias -> (bf16 panel [ceil32(Cout)][ceil16(Cin)], f32 bias [ceil32(Cout)])."""
    cout, cin = w.shape
    cop, cip = -(-cout // 32) * 32, -(-cin // 16) * 16
    wp = torch.zeros(cop, cip, dtype=BF16, device=w.device)
    wp[:cout, :cin] = w.to(BF16)
    bp = torch.zeros(cop, dtype=F32, device=w.device)
    bp[:cout] = bias
    return wp, bp


def mb_pointwise(x: torch.Tensor, wp: torch.Tensor, bp: torch.Tensor, cout: int, act: int, gate=None, res=None) -> torch.Tensor:
    """y (N,H,W,Cout) bf16 = act(W (x * gate) + bias) (+ res) for x (N,H,W,Cin) bf16."""
    if x.dim() != 4 or x.dtype != BF16:
        raise ValueError("mb_pointwise: expected (N,H,W,C) bf16")
    Nn, H, W, cin = x.shape
    if tuple(wp.shape) != (-(-cout // 32) * 32, -(-cin // 16) * 16) or wp.dtype != BF16 or bp.numel() != wp.shape[0]:
        raise ValueError(f"mb_pointwise: weight panel {tuple(wp.shape)} does not fit Cout={cout}, Cin={cin}")
    if gate is not None:
        _chk4(gate, (Nn, cin), "gate")
    y = torch.empty(Nn, H, W, cout, dtype=BF16, device=x.device)
    if res is not None and (tuple(res.shape) != tuple(y.shape) or res.dtype != BF16):
        raise ValueError("mb_pointwise: residual must match the output")
    check(lib().fdet_mb_pointwise(ptr(x, BF16), ptr(wp, BF16), ptr(bp), ptr(gate), ptr(res, BF16), ptr(y, BF16), Nn, H * W, cin,
                                  cout, act, stream()), "fdet_mb_pointwise")
    return y


def mb_head_pack(w: torch.Tensor) -> torch.Tensor:
    """Conv2d(C,5,3,p1) weight (5,C,3,3) f32 -> bf16 (2,64,C): row tap*5 + ch (rows 45..63 zero), split into
    hi = bf16(w), lo = bf16(w - hi)."""
    C = w.shape[1]
    wt = torch.zeros(64, C, dtype=F32, device=w.device)
    wt[:45] = w.float().permute(2, 3, 0, 1).reshape(45, C)       # (ky, kx, ch) -> tap*5 + ch
    hi = wt.to(BF16)
    lo = (wt - hi.float()).to(BF16)
    return torch.stack([hi, lo]).contiguous()


def mb_head(f: torch.Tensor, w2: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """f (N,S,S,C) bf16 -> sigmoid(Conv2d(C,5,3,p1)) as (N,5,S,S) f32; w2 = mb_head_pack(weight)."""
    if f.dim() != 4 or f.dtype != BF16 or f.shape[1] != f.shape[2] or f.shape[3] % 16:
        raise ValueError("mb_head: expected (N,S,S,C) bf16 with C % 16 == 0")
    Nn, S, _, C = f.shape
    if tuple(w2.shape) != (2, 64, C) or w2.dtype != BF16:
        raise ValueError(f"mb_head: expected the packed weight (2,64,{C}) bf16, got {tuple(w2.shape)} {w2.dtype}")
    _chk4(bias, (5,), "head bias")
    y = torch.empty(Nn, 5, S, S, dtype=F32, device=f.device)
    nb = int(lib().fdet_mb_head_ws_bytes(Nn, S))
    ws = torch.empty(nb // 4, dtype=F32, device=f.device)
    check(lib().fdet_mb_head(ptr(f, BF16), ptr(w2, BF16), ptr(bias), ptr(y), ptr(ws), nb, Nn, S, C, stream()), "fdet_mb_head")
    return y
