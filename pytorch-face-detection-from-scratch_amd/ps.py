"""Engine-private pre-split (PS) activations of the bf16x3 conv stack (csrc/fdet_ps.h, include/fdet.h).

A `PsTensor` owns ONE zero-filled allocation (two guard images included) and hands the C-ABI the pointer of
image 0.  Producers only ever write real elements, so the halo slots / rows / guard images stay zero for the life
of the buffer -- which is why the engine keeps these buffers across steps instead of re-allocating them.

No counterpart in the reference (its tensors are fp32 NCHW, models/PoolResnet.py:33-43): fp32 NCHW remains the
format at every boundary that mirrors a reference interface.
"""
from __future__ import annotations

import functools

import torch

from ._native import FdetError, check, lib, ptr, stream

F32 = torch.float32


@functools.lru_cache(maxsize=None)
def strips_of(W: int):
    """(S, Ws): column strips of a map of width W (csrc/fdet_ps.h) -- (1, W) for maps of up to 63 columns, (0, 0) when the
    width has no layout (odd and wider than 63)."""
    S = int(lib().fdet_ps_strips(int(W)))
    if S <= 1:
        return S, (W if S == 1 else 0)
    return S, ((W + S - 1) // S + 1) & ~1


class PsTensor:
    """bf16 hi|lo planes of a (N,C,H,W) feature map, 8 channels innermost, zero halos (see csrc/fdet_ps.h).  Maps wider than
    63 columns are kept as column strips (`strips` > 1): the C-ABI plans them from the full width, the caller's part is
    `halo_exchange` (below)."""

    def __init__(self, N: int, C: int, H: int, W: int, device):
        nbytes = int(lib().fdet_ps_bytes(N, C, H, W))
        if nbytes == 0:
            raise FdetError(f"PsTensor: unsupported shape ({N},{C},{H},{W}) (C % 8 == 0; W <= 63 or even)")
        self.shape = (N, C, H, W)
        self.strips = strips_of(W)[0]
        self.buf = torch.zeros(nbytes // 4, dtype=F32, device=device)        # zeroed ONCE
        self._off = int(lib().fdet_ps_image0_offset(N, C, H, W))

    @property
    def data(self) -> int:
        """Device pointer of image 0 (what every fdet_*_ps_* entry point takes)."""
        return ptr(self.buf) + self._off

    @property
    def device(self):
        return self.buf.device

    @classmethod
    def from_f32(cls, x: torch.Tensor, out: "PsTensor" = None) -> "PsTensor":
        if x.dim() != 4 or x.dtype != F32 or not x.is_contiguous():
            raise TypeError("PsTensor.from_f32: expected a contiguous fp32 (N,C,H,W) tensor")
        N, C, H, W = x.shape
        t = out if out is not None else cls(N, C, H, W, x.device)
        if t.shape != (N, C, H, W):
            raise ValueError(f"PsTensor.from_f32: buffer is {t.shape}, tensor is {tuple(x.shape)}")
        check(lib().fdet_ps_from_f32(ptr(x), t.data, N, C, H, W, stream()), "fdet_ps_from_f32")
        return t

    def to_f32(self, out: torch.Tensor = None) -> torch.Tensor:
        N, C, H, W = self.shape
        y = out if out is not None else torch.empty(N, C, H, W, dtype=F32, device=self.device)
        if tuple(y.shape) != self.shape:
            raise ValueError("PsTensor.to_f32: output shape mismatch")
        check(lib().fdet_ps_to_f32(self.data, ptr(y), N, C, H, W, stream()), "fdet_ps_to_f32")
        return y


def halo_exchange(t: PsTensor, zero_only: bool = False, p16: bool = False) -> None:
    """Strip tensors (a no-op otherwise): copy the neighbour strips' edge columns into the halo slots (after a producer that
    wrote real elements only, before a 3x3 conv reads `t`), or -- zero_only -- clear them (before `t` is the dz operand of the
    weight gradient)."""
    if t.strips <= 1:
        return
    N, C, H, W = t.shape
    check(lib().fdet_ps_halo_exchange(t.data, N, C, H, W, int(zero_only), int(p16), stream()), "fdet_ps_halo_exchange")


def route8_shape(N: int, C: int, H: int, W: int):
    """Shape of the routing bytes of a pooled block whose conv maps are (H, W): per strip-image, channel-innermost."""
    S, Ws = strips_of(W)
    return (N * S, C // 8, H // 2, Ws // 2, 8)


def _same(a: PsTensor, shape, name):
    if not isinstance(a, PsTensor) or a.shape != tuple(shape):
        raise ValueError(f"{name}: expected a PsTensor of shape {tuple(shape)}, got {getattr(a, 'shape', type(a))}")


def _fn(name: str, p16: bool):
    """The C-ABI entry point `name`, or its one-pass `_p16` twin (precision16: bf16 hi planes only, include/fdet.h)."""
    return getattr(lib(), name + "_p16" if p16 else name)


def conv3x3_ps_fwd(x: PsTensor, wpk: torch.Tensor, bias: torch.Tensor, y: PsTensor, slope: float = 0.2, p16: bool = False) -> None:
    """y = LeakyReLU(conv3x3(x) + bias) on PS tensors; wpk: forward panel of pack_conv3x3_weights(x3=True)."""
    N, cin, H, W = x.shape
    cout = int(bias.shape[0])
    _same(y, (N, cout, H, W), "conv3x3_ps_fwd: y")
    if wpk.numel() != cout * cin * 9:
        raise ValueError("conv3x3_ps_fwd: packed weight size does not match (Cout,Cin)")
    check(_fn("fdet_conv3x3_ps_fwd", p16)(x.data, ptr(wpk), ptr(bias), y.data, N, cin, cout, H, W, float(slope), stream()),
          "fdet_conv3x3_ps_fwd")


def conv3x3_ps_dgrad_act(dz: PsTensor, wpk_bwd: torch.Tensor, act: PsTensor, dx: PsTensor, slope: float = 0.2, p16: bool = False) -> None:
    """dx = conv3x3^T(dz) * LeakyReLU'(act) on PS tensors; wpk_bwd: backward panel."""
    N, cout, H, W = dz.shape
    cin = dx.shape[1]
    _same(dx, (N, cin, H, W), "conv3x3_ps_dgrad_act: dx")
    _same(act, (N, cin, H, W), "conv3x3_ps_dgrad_act: act")
    if wpk_bwd.numel() != cout * cin * 9:
        raise ValueError("conv3x3_ps_dgrad_act: packed weight size does not match (Cout,Cin)")
    check(_fn("fdet_conv3x3_ps_dgrad_act", p16)(dz.data, ptr(wpk_bwd), act.data, dx.data, N, cin, cout, H, W, float(slope),
                                                stream()), "fdet_conv3x3_ps_dgrad_act")


def conv3x3_wgrad_ps_ws_bytes(L: int, N: int, C: int, H: int, W: int) -> int:
    return int(lib().fdet_conv3x3_wgrad_ps_ws_bytes(L, N, C, H, W))


def conv3x3_wgrad_ps_batched(xs, dzs, dWs, dbs, ws: torch.Tensor, p16: bool = False) -> None:
    """dWs[l] (64,64,3,3), dbs[l] (64,) = weight / bias gradients of L same-shape layers from PS operands."""
    import ctypes
    L = len(xs)
    N, C, H, W = xs[0].shape
    for x, dz, dW, db in zip(xs, dzs, dWs, dbs):
        _same(x, (N, C, H, W), "conv3x3_wgrad_ps: x")
        _same(dz, (N, C, H, W), "conv3x3_wgrad_ps: dz")
        if tuple(dW.shape) != (C, C, 3, 3) or tuple(db.shape) != (C,):
            raise ValueError("conv3x3_wgrad_ps: dW / db shapes")
    arr = ctypes.c_void_p * L
    check(_fn("fdet_conv3x3_wgrad_ps_batched", p16)(arr(*[x.data for x in xs]), arr(*[z.data for z in dzs]),
                                                    arr(*[ptr(t) for t in dWs]), arr(*[ptr(t) for t in dbs]), L, N, C, H, W,
                                                    ptr(ws), ws.numel() * 4, stream()), "fdet_conv3x3_wgrad_ps_batched")


def route8_like(N: int, C: int, H: int, W: int, device) -> torch.Tensor:
    """Routing bytes of a pooled block whose conv maps are (H, W): uint8 [N, C/8, H/2, W/2, 8]."""
    return torch.empty(*route8_shape(N, C, H, W), dtype=torch.uint8, device=device)


def conv3x3_ps_fwd_pool(x: PsTensor, wpk, bias, skip: PsTensor, drop_scale, pool_ps: PsTensor = None,
                        pool_f32: torch.Tensor = None, route8: torch.Tensor = None, slope: float = 0.2, p16: bool = False) -> None:
    """pooled = maxpool2x2(lrelu(conv(x)+bias) * drop_scale + skip) -> pool_ps and / or pool_f32; route8: routing bytes."""
    N, cin, H, W = x.shape
    cout = int(bias.shape[0])
    _same(skip, (N, cout, H, W), "conv3x3_ps_fwd_pool: skip")
    if pool_ps is not None:
        _same(pool_ps, (N, cout, H // 2, W // 2), "conv3x3_ps_fwd_pool: pool_ps")
    if pool_f32 is not None and tuple(pool_f32.shape) != (N, cout, H // 2, W // 2):
        raise ValueError("conv3x3_ps_fwd_pool: pool_f32 shape")
    if route8 is not None and tuple(route8.shape) != route8_shape(N, cout, H, W):
        raise ValueError("conv3x3_ps_fwd_pool: route8 shape")
    if pool_ps is not None and x.strips > 1:
        raise ValueError("conv3x3_ps_fwd_pool: a strip map writes its pooled output as fp32 NCHW (pool_f32)")
    if drop_scale is not None and tuple(drop_scale.shape) != (N, cout):
        raise ValueError("conv3x3_ps_fwd_pool: drop_scale shape")
    if wpk.numel() != cout * cin * 9:
        raise ValueError("conv3x3_ps_fwd_pool: packed weight size does not match (Cout,Cin)")
    check(_fn("fdet_conv3x3_ps_fwd_pool", p16)(x.data, ptr(wpk), ptr(bias), skip.data, ptr(drop_scale),
                                               pool_ps.data if pool_ps is not None else None, ptr(pool_f32),
                                               ptr(route8, torch.uint8), N, cin, cout, H, W, float(slope), stream()),
          "fdet_conv3x3_ps_fwd_pool")


def pool_route_bwd_ps(dout_pooled: torch.Tensor, route8: torch.Tensor, drop_scale, dz2: PsTensor, slope: float = 0.2,
                      p16: bool = False) -> None:
    """dz2 (PS) = unpool(dout_pooled) * drop_scale * lrelu'(c) from the routing bytes."""
    N, C, H, W = dz2.shape
    if tuple(dout_pooled.shape) != (N, C, H // 2, W // 2) or tuple(route8.shape) != route8_shape(N, C, H, W):
        raise ValueError("pool_route_bwd_ps: shapes")
    check(_fn("fdet_pool_route_bwd_ps", p16)(ptr(dout_pooled), ptr(route8, torch.uint8), ptr(drop_scale), dz2.data, N, C, H, W,
                                             float(slope), stream()), "fdet_pool_route_bwd_ps")


def conv3x3_ps_dgrad_unpool(dz: PsTensor, wpk_bwd, dout_pooled: torch.Tensor, route8: torch.Tensor, dx: torch.Tensor,
                            slope: float = 0.2, p16: bool = False) -> None:
    """dx (fp32 NCHW) = conv3x3^T(dz) + unpool(dout_pooled) through the routing bytes."""
    N, cout, H, W = dz.shape
    cin = dx.shape[1]
    if tuple(dx.shape) != (N, cin, H, W) or tuple(dout_pooled.shape) != (N, cin, H // 2, W // 2) or \
            tuple(route8.shape) != route8_shape(N, cin, H, W):
        raise ValueError("conv3x3_ps_dgrad_unpool: shapes")
    if wpk_bwd.numel() != cout * cin * 9:
        raise ValueError("conv3x3_ps_dgrad_unpool: packed weight size does not match (Cout,Cin)")
    check(_fn("fdet_conv3x3_ps_dgrad_unpool", p16)(dz.data, ptr(wpk_bwd), ptr(dout_pooled), ptr(route8, torch.uint8), ptr(dx),
                                                   N, cin, cout, H, W, float(slope), stream()), "fdet_conv3x3_ps_dgrad_unpool")


def stem_fwd_ps(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, y: PsTensor, k: int, stride: int, pad: int,
                p16: bool = False) -> None:
    """PoolResnet stem (3 -> 64 channels, k10 s8 p2) or Resnet stem (k3 s2 p1) with a PS output.  A uint8 `x` (inference:
    the frames themselves, PoolResnet stem only) is normalised by 255 inside the kernel (fdet_stem_fwd_ps_u8)."""
    N, cin, H, W = x.shape
    F_ = int(w.shape[0])
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    _same(y, (N, F_, Ho, Wo), "stem_fwd_ps: y")
    if x.dtype == torch.uint8:
        if not x.is_contiguous():
            raise ValueError("stem_fwd_ps: uint8 frames must be contiguous")
        check(lib().fdet_stem_fwd_ps_u8(ptr(x, torch.uint8), ptr(w), ptr(bias), y.data, N, cin, F_, H, W, k, stride, pad, int(p16), stream()),
              "fdet_stem_fwd_ps_u8")
        return
    check(_fn("fdet_stem_fwd_ps", p16)(ptr(x), ptr(w), ptr(bias), y.data, N, cin, F_, H, W, k, stride, pad, stream()), "fdet_stem_fwd_ps")


def _ps_ptr_array(ts):
    """HOST array of PS image-0 pointers (None entries -> NULL); None -> NULL array."""
    import ctypes
    if ts is None:
        return None
    return (ctypes.c_void_p * len(ts))(*[(t.data if t is not None else None) for t in ts])


def block_chain_fwd_ps(x, wpk1, b1, wpk2, b2, scales, a_ps, c_ps, out_ps, out_last: torch.Tensor, slope: float = 0.2,
                       p16: bool = False) -> None:
    """The LDS-resident residual-block chain (fdet_block_chain_fwd_ps) keeping a_k / c_k / block outputs as PS tensors.
    x: PsTensor or fp32 NCHW; a_ps, c_ps: lists of nblocks PsTensors (or None: not kept); out_ps: nblocks-1 PsTensors
    (or None); out_last: fp32 NCHW output of the last block.  c_ps[k] only receives its hi plane (its signs)."""
    from .hotpath import _ptr_array
    nb = len(wpk1)
    N, C, H, W = out_last.shape
    x_is_ps = isinstance(x, PsTensor)
    if x_is_ps:
        _same(x, (N, C, H, W), "block_chain_fwd_ps: x")
    elif tuple(x.shape) != (N, C, H, W) or x.dtype != F32 or not x.is_contiguous():
        raise ValueError("block_chain_fwd_ps: x must be a PsTensor or a contiguous fp32 tensor of the output's shape")
    if not (len(b1) == len(wpk2) == len(b2) == nb) or (scales is not None and len(scales) != nb):
        raise ValueError("block_chain_fwd_ps: inconsistent per-block lists")
    for lst, n_, nm in ((a_ps, nb, "a"), (c_ps, nb, "c"), (out_ps, nb - 1, "out")):
        if lst is not None:
            if len(lst) != n_:
                raise ValueError(f"block_chain_fwd_ps: {nm} needs {n_} entries")
            for t in lst:
                if t is not None:
                    _same(t, (N, C, H, W), "block_chain_fwd_ps: " + nm)
    check(_fn("fdet_block_chain_fwd_ps", p16)(x.data if x_is_ps else ptr(x), int(x_is_ps), _ptr_array(wpk1), _ptr_array(b1),
                                        _ptr_array(wpk2), _ptr_array(b2), _ptr_array(scales), _ps_ptr_array(a_ps),
                                        _ps_ptr_array(c_ps), _ps_ptr_array(out_ps), ptr(out_last), nb, N, C, H, W,
                                        float(slope), stream()), "fdet_block_chain_fwd_ps")


def block_chain_bwd_ps(dout: torch.Tensor, wpk1b, wpk2b, scales, a_ps, c_ps, dz1_ps, dz2_ps, dx: torch.Tensor,
                       slope: float = 0.2, p16: bool = False) -> None:
    """Data-gradient chain of the same blocks (fdet_block_chain_bwd_ps): fills dz1_ps[k], dz2_ps[k] (PS) and dx (fp32)."""
    from .hotpath import _ptr_array
    nb = len(wpk1b)
    N, C, H, W = dout.shape
    if not (len(wpk2b) == len(a_ps) == len(c_ps) == len(dz1_ps) == len(dz2_ps) == nb):
        raise ValueError("block_chain_bwd_ps: inconsistent per-block lists")
    for lst, nm in ((a_ps, "a"), (c_ps, "c"), (dz1_ps, "dz1"), (dz2_ps, "dz2")):
        for t in lst:
            _same(t, (N, C, H, W), "block_chain_bwd_ps: " + nm)
    if tuple(dx.shape) != (N, C, H, W):
        raise ValueError("block_chain_bwd_ps: dx shape mismatch")
    check(_fn("fdet_block_chain_bwd_ps", p16)(ptr(dout), _ptr_array(wpk1b), _ptr_array(wpk2b), _ptr_array(scales),
                                        _ps_ptr_array(a_ps), _ps_ptr_array(c_ps), _ps_ptr_array(dz1_ps),
                                        _ps_ptr_array(dz2_ps), ptr(dx), nb, N, C, H, W, float(slope), stream()),
          "fdet_block_chain_bwd_ps")
