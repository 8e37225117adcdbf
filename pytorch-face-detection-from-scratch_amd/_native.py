"""ctypes binding of libfdet_hip.so (the C-ABI in include/fdet.h).

This is the only way the package reaches the GPU: every public op in this package ends in
one of the calls below.  There is NO fallback -- a missing library, a CPU tensor or a
non-contiguous/ non-fp32 buffer raises immediately (FdetError / TypeError).
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
from typing import Optional

import torch  # imported BEFORE the CDLL so that libamdhip64.so.7 resolves to torch's runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDET_LIB_PATH") or os.path.join(_HERE, "lib", "libfdet_hip.so")   # override: development builds (tools/probe)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fdet.h")


class FdetError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into lib/libfdet_hip.so (hipcc cross-compiles
    without a GPU).  Returns the library path."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-8000:])
    if r.returncode != 0:
        raise FdetError("building libfdet_hip.so failed (see compiler output above)")
    return LIB_PATH


_c = ctypes
_P, _I, _F, _D, _SZ, _U64 = _c.c_void_p, _c.c_int, _c.c_float, _c.c_double, _c.c_size_t, _c.c_uint64

# name -> (restype, argtypes); mirrors include/fdet.h declaration by declaration
SIGNATURES = {
    "fdet_version": (_I, []),
    "fdet_last_error": (_c.c_char_p, []),
    "fdet_encode_targets": (_I, [_P, _P, _I, _I, _F, _F, _P, _P]),
    "fdet_yolo_loss_fwd_bwd": (_I, [_P, _P, _I, _I, _P, _P, _P, _F, _P]),
    "fdet_decode": (_I, [_P, _I, _I, _F, _F, _F, _P, _P, _P, _P]),
    "fdet_nms": (_I, [_P, _P, _P, _I, _I, _D, _P, _P, _P]),
    "fdet_reduce_bounding_boxes": (_I, [_P, _I, _I, _F, _D, _F, _F, _P, _P, _P]),
    "fdet_step_metrics": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "fdet_ssd_num_priors": (_I, [_P, _I]),
    "fdet_ssd_encode_targets": (_I, [_P, _P, _I, _P, _I, _F, _F, _P, _P]),
    "fdet_ssd_loss_ws_bytes": (_SZ, [_I]),
    "fdet_ssd_loss_fwd_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "fdet_ssd_loss_parts": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "fdet_ssd_loss_finish": (_I, [_P, _P, _P, _SZ, _P, _SZ, _P]),
    "fdet_ssd_reduce_bounding_boxes": (_I, [_P, _I, _P, _I, _I, _F, _D, _F, _F, _P, _P, _P]),
    "fdet_ssd_reduce_bounding_boxes_priors": (_I, [_P, _I, _P, _I, _I, _P, _F, _D, _F, _F, _P, _P, _P]),
    "fdet_ssd_head_pack_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "fdet_ssd_head_pack_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "fdet_u8_to_f32_norm": (_I, [_P, _P, _SZ, _P]),
    "fdet_resize_bilinear_u8_norm": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_resize_bilinear_f32_norm": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_adam_step": (_I, [_P, _P, _P, _P, _SZ, _I, _D, _D, _D, _D, _F, _P]),
    "fdet_pack_conv3x3_weights": (_I, [_P, _I, _I, _P, _P, _P]),
    "fdet_conv3x3_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_dgrad": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_pack_conv3x3_weights_bf16x3": (_I, [_P, _I, _I, _P, _P, _P]),
    "fdet_pack_conv3x3_weights_bf16x3_batched": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "fdet_conv3x3_fwd_bf16x3": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_dgrad_bf16x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_fwd_pool_bf16x3": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_dgrad_unpool_bf16x3": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_pool_route_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_pool_fusion_ok": (_I, [_I, _I, _I, _I, _I]),
    "fdet_ps_bytes": (_SZ, [_I, _I, _I, _I]),
    "fdet_ps_image0_offset": (_SZ, [_I, _I, _I, _I]),
    "fdet_ps_from_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fdet_ps_to_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "fdet_ps_strips": (_I, [_I]),
    "fdet_ps_halo_exchange": (_I, [_P, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_conv3x3_ps_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_ps_dgrad_act": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_ps_fwd_pool": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_pool_route_bwd_ps": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_ps_dgrad_unpool": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_stem_fwd_ps": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_conv3x3_wgrad_ps_ws_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "fdet_conv3x3_wgrad_ps_batched": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "fdet_pointwise_packed_bytes": (_SZ, [_I, _I]),
    "fdet_pack_pointwise_weights_bf16x3": (_I, [_P, _I, _I, _P, _P, _P]),
    "fdet_pointwise_fwd_bf16x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "fdet_pointwise_dgrad_bf16x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fdet_pointwise_wgrad_ws_bytes": (_SZ, [_I, _I, _I, _I]),
    "fdet_pointwise_wgrad_bf16x3": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "fdet_mb_stem": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "fdet_mb_depthwise": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_mb_depthwise_pool_slots": (_I, [_I, _I, _I, _I, _I, _I]),
    "fdet_mb_se_gate": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "fdet_mb_pointwise": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fdet_mb_head_ws_bytes": (_SZ, [_I, _I]),
    "fdet_mb_head": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "fdet_conv3x3_wgrad_ws_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "fdet_conv3x3_wgrad": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _P]),
    "fdet_conv3x3_wgrad_bf16x3_ws_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "fdet_conv3x3_wgrad_bf16x3": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _P]),
    "fdet_conv3x3_wgrad_bf16x3_batched_ws_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "fdet_conv3x3_wgrad_bf16x3_batched": (_I, [_P, _P, _P, _P, _I, _P, _SZ, _I, _I, _I, _I, _I, _P]),
    "fdet_block_chain_supported": (_I, [_I, _I, _I]),
    "fdet_block_chain_fwd_bf16x3": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_block_chain_bwd_bf16x3": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_block_chain_fwd_ps": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_block_chain_bwd_ps": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_block_tail_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "fdet_block_tail_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_stem_ws_bytes": (_SZ, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "fdet_stem_fwd": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_stem_fwd_bf16x3": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_stem_wgrad": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_stem_wgrad_bf16x3": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_head_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_head_bwd_ws_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "fdet_head_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_conv3x3_ps_fwd_p16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_ps_dgrad_act_p16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_ps_fwd_pool_p16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_conv3x3_ps_dgrad_unpool_p16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_pool_route_bwd_ps_p16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "fdet_stem_fwd_ps_p16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_stem_fwd_ps_u8": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_stem_wgrad_bf16": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_conv3x3_wgrad_ps_batched_p16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "fdet_block_chain_fwd_ps_p16": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_block_chain_bwd_ps_p16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "fdet_mbt_stem_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "fdet_mbt_taps_ws_bytes": (_SZ, [_I, _I]),
    "fdet_mbt_stem_wgrad": (_I, [_P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "fdet_mbt_dw_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_mbt_dw_bwd": (_I, [_P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_mbt_bn_ws_bytes": (_SZ, [_I]),
    "fdet_mbt_bn_fwd": (_I, [_P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "fdet_mbt_bn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "fdet_mbt_se_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "fdet_mbt_se_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "fdet_head_loss_fused_supported": (_I, [_I, _I, _I, _I, _I]),
    "fdet_head_loss_fused_ws_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "fdet_head_loss_fused": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _I, _I, _P]),
    "fdet_dropout_scales": (_I, [_P, _SZ, _F, _U64, _U64, _P]),
    "fdet_dropout_scales_layers": (_I, [_P, _I, _I, _P, _P, _U64, _U64, _U64, _P]),
}


def header_symbols() -> list:
    """Every function name declared in include/fdet.h (used by the symbol-export test)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fdet_[a-z0-9_]+)\s*\(", txt)))


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """The loaded library; raises FdetError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdetError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.fdet_version() != 100:
            raise FdetError(f"libfdet_hip.so version {L.fdet_version()} != 100")
        _lib = L
    return _lib


def hip_runtimes_mapped() -> list:
    out = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                out.add(line.split()[-1])
    return sorted(out)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().fdet_last_error().decode(errors="replace")
        raise FdetError(f"{what} failed with code {rc}: {msg}")


def ptr(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    """Device pointer of a dense GPU tensor (None -> NULL).  Loud on anything else."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise FdetError("libfdet_hip.so works on GPU memory only; got a CPU tensor (no CPU fallback exists)")
    if t.dtype != dtype:
        raise TypeError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise TypeError("expected a contiguous tensor")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream
