"""MobileNetV3-small backbone engine (BASELINE.json config 5; models/MobilenetV3Backbone.py:11-60), inference, bf16.

The reference builds the backbone with `timm.create_model("tf_mobilenetv3_small_100")` (third party, absent here); the
architecture below is read off the parameter shapes and the embedded source text of the shipped archive
(saved_models/official/MobilenetV3Backbone/medium_model_15x15_480.pth).

Layout in HBM: activations NHWC bf16 (channel vectors of 8 = 16-byte accesses for the depthwise and pointwise kernels);
BatchNorm (eval, eps 1e-3) folded into fp32 conv weights, pointwise weights then rounded to bf16 panels
[ceil32(Cout)][ceil16(Cin)].  Every layer reads its input once and writes its output once; the SqueezeExcite pooling is a
by-product of the depthwise kernel and the gate is applied while the projection GEMM loads its operand.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import hotpath as hp

BN_EPS = 1e-3
# (kind, cin, cexp, cout, kernel, stride, act, se_reduce or 0): ds = DepthwiseSeparable, ir = InvertedResidual
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
STAGES = [[0], [1, 2], [3, 4, 5], [6, 7], [8, 9, 10]]          # block indices per timm stage; stage 5 = ConvBnAct 96->576
FEATURES = 576


def block_prefix(b: int) -> str:
    for s, idx in enumerate(STAGES):
        if b in idx:
            return f"feature_extractor.3.{s}.{idx.index(b)}"
    raise IndexError(b)


def _fold(w: torch.Tensor, P: Dict[str, torch.Tensor], bn: str) -> Tuple[torch.Tensor, torch.Tensor]:
    scale = P[bn + ".weight"].float() / torch.sqrt(P[bn + ".running_var"].float() + BN_EPS)
    return w.float() * scale.view(-1, 1, 1, 1), P[bn + ".bias"].float() - P[bn + ".running_mean"].float() * scale


class MobileNetStack:
    """`pack(state)` once per set of weights, then `forward(x)`: (N,3,H,W) f32 in [0,1] (or uint8, /255 fused into the
    stem) -> (N,5,H/32,W/32) f32 sigmoid maps."""

    def __init__(self):
        self.packed = None
        self.timer = None                    # optional callable(label, bytes, flops) -> context manager (tools/run_config5.py)

    # ------------------------------------------------------------------ weights
    def pack(self, P: Dict[str, torch.Tensor]) -> None:
        dev = P["out.weight"].device
        if dev.type != "cuda":
            raise hp.N.FdetError("the MobileNet stack runs on the GPU only (no CPU fallback): move the model to cuda")
        L: List[dict] = []
        w, b = _fold(P["feature_extractor.0.weight"], P, "feature_extractor.1")
        stem = (w.reshape(16, 27).contiguous(), b.contiguous())
        for bi, (kind, ci, ce, co, k, s, act, se) in enumerate(BLOCKS):
            p = block_prefix(bi)
            e: dict = {"kind": kind, "ci": ci, "ce": ce, "co": co, "k": k, "s": s, "act": hp.MB_ACT[act], "se": se}
            if kind == "ir":
                w, b = _fold(P[p + ".conv_pw.weight"], P, p + ".bn1")
                e["pw"] = hp.mb_pointwise_pack(w.reshape(ce, ci), b)
            w, b = _fold(P[p + ".conv_dw.weight"], P, p + (".bn2" if kind == "ir" else ".bn1"))
            e["dw"] = (w.reshape(ce, k * k).t().contiguous(), b.contiguous())
            if se:
                e["se_w"] = (P[p + ".se.conv_reduce.weight"].float().reshape(se, ce).contiguous(),
                             P[p + ".se.conv_reduce.bias"].float().contiguous(),
                             P[p + ".se.conv_expand.weight"].float().reshape(ce, se).contiguous(),
                             P[p + ".se.conv_expand.bias"].float().contiguous())
            last = ".conv_pwl.weight" if kind == "ir" else ".conv_pw.weight"
            w, b = _fold(P[p + last], P, p + (".bn3" if kind == "ir" else ".bn2"))
            e["pwl"] = hp.mb_pointwise_pack(w.reshape(co, ce), b)
            L.append(e)
        w, b = _fold(P["feature_extractor.3.5.0.conv.weight"], P, "feature_extractor.3.5.0.bn1")
        final = hp.mb_pointwise_pack(w.reshape(FEATURES, 96), b)
        head = (hp.mb_head_pack(P["out.weight"]), P["out.bias"].float().contiguous())
        self.packed = (stem, L, final, head)

    # ------------------------------------------------------------------ forward
    def _t(self, label, nbytes, flops):
        return self.timer(label, nbytes, flops) if self.timer is not None else _NULL

    def stem(self, x: torch.Tensor) -> torch.Tensor:
        """(N,3,H,W) f32 / uint8 -> (N,H/2,W/2,16) bf16."""
        if self.packed is None:
            raise hp.N.FdetError("MobileNetStack.forward before pack()")
        if not x.is_cuda:
            raise hp.N.FdetError("the MobileNet stack runs on the GPU only (no CPU fallback): move the input to cuda")
        x = x.contiguous()
        with self._t("stem", x.numel() * x.element_size() + x.shape[0] * (x.shape[2] // 2) * (x.shape[3] // 2) * 32, 0):
            return hp.mb_stem(x, *self.packed[0])

    def block(self, bi: int, h: torch.Tensor) -> torch.Tensor:
        """Block `bi` of BLOCKS on an NHWC bf16 activation."""
        e = self.packed[1][bi]
        skip = h
        ce, co = e["ce"], e["co"]
        if e["kind"] == "ir":
            with self._t(f"b{bi}.pw", h.numel() * 2 + h.numel() // e["ci"] * ce * 2, 2 * h.numel() * ce):
                h = hp.mb_pointwise(h, *e["pw"], ce, e["act"])
        Ho, Wo = -(-h.shape[1] // e["s"]), -(-h.shape[2] // e["s"])
        with self._t(f"b{bi}.dw{e['k']}s{e['s']}", h.numel() * 2 + h.shape[0] * Ho * Wo * ce * 2, 2 * h.shape[0] * Ho * Wo * ce * e["k"] ** 2):
            h, pool = hp.mb_depthwise(h, *e["dw"], e["k"], e["s"], e["act"], bool(e["se"]))
        gate = None
        if e["se"]:
            with self._t(f"b{bi}.se", 0, 0):
                gate = hp.mb_se_gate(pool, h.shape[1] * h.shape[2], *e["se_w"])
        res = skip if (e["s"] == 1 and e["ci"] == co) else None
        with self._t(f"b{bi}.pwl", h.numel() * 2 + h.numel() // ce * co * 2 * (2 if res is not None else 1), 2 * h.numel() * co):
            return hp.mb_pointwise(h, *e["pwl"], co, 0, gate, res)

    def final(self, h: torch.Tensor) -> torch.Tensor:
        with self._t("final.pw", h.numel() * 2 + h.numel() // 96 * FEATURES * 2, 2 * h.numel() * FEATURES):
            return hp.mb_pointwise(h, *self.packed[2], FEATURES, 2)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        """-> (N,H/32,W/32,576) bf16, NHWC."""
        h = self.stem(x)
        for bi in range(len(BLOCKS)):
            h = self.block(bi, h)
        return self.final(h)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        f = self.features(x)
        with self._t("head", f.numel() * 2, 2 * f.numel() * 45):
            return hp.mb_head(f, *self.packed[3])


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _Null()
