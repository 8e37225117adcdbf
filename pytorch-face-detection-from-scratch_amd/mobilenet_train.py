"""Training engine of the MobileNetV3-small backbone model (round 4): forward in TRAINING mode
(BatchNorm batch statistics, running statistics updated) and the hand-written backward, as a fixed sequence of HIP
kernel launches -- what `ModelMeta.training_step` + `loss.backward()` need from `MobilenetV3Backbone`
(models/MobilenetV3Backbone.py:49-60; timm tf_mobilenetv3_small_100 features + Conv2d(576,5,3,p1) + sigmoid).

fp32 NCHW (the reference's layout); 1x1 convs = fdet_pointwise_*_bf16x3, head = fdet_head_fwd / fdet_head_bwd, the rest =
csrc/fdet_mobilenet_train.hip.  The inference engine (mobilenetstack.py: bf16 NHWC, BatchNorm folded) is unchanged.
Correctness first: BASELINE config 5 is an inference run, this path exists so that the reference's training surface works on
this backbone too.  PARITY UNPINNED (timm absent); checked against torch autograd on oracle/mobilenet_oracle.py.

Data parallelism (SURVEY 8f-3's SyncBN question): statistics are PER RANK -- exactly what the reference does on one GPU per
process under Lightning DDP without `sync_batchnorm=True` (train_model.py:47-53 does not set it).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import hotpath as hp
from ._native import check, lib, ptr, stream
from .mobilenetstack import BLOCKS, BN_EPS, FEATURES, block_prefix

F32 = torch.float32
ACT = {"none": 0, "relu": 1, "hswish": 2}
BN_MOMENTUM = 0.01          # timm tf_ models: bn_momentum = 1 - 0.99 (the mirror's nn.BatchNorm2d(momentum=0.01))


def _bn_ws(C: int, dev) -> torch.Tensor:
    return torch.empty(int(lib().fdet_mbt_bn_ws_bytes(C)) // 4 + 4, dtype=F32, device=dev)


def bn_fwd(z, P, bn, act: int, residual=None, update_running: bool = True):
    """act(BatchNorm_train(z)) (+ residual) -> (y, (mean, invstd)); updates P[bn.running_*] and num_batches_tracked."""
    N, C = z.shape[0], z.shape[1]
    Pn = z.numel() // (N * C)
    y = torch.empty_like(z)
    mean = torch.empty(C, dtype=F32, device=z.device)
    invstd = torch.empty(C, dtype=F32, device=z.device)
    ws = _bn_ws(C, z.device)
    rm = P[bn + ".running_mean"] if update_running else None
    rv = P[bn + ".running_var"] if update_running else None
    check(lib().fdet_mbt_bn_fwd(ptr(z), ptr(P[bn + ".weight"]), ptr(P[bn + ".bias"]), ptr(rm), ptr(rv), BN_MOMENTUM, BN_EPS, ptr(mean),
                                ptr(invstd), ptr(residual), ptr(y), ptr(ws), ws.numel() * 4, N, C, Pn, act, stream()), "fdet_mbt_bn_fwd")
    if update_running and (bn + ".num_batches_tracked") in P:
        P[bn + ".num_batches_tracked"] += 1
    return y, (mean, invstd)


def bn_bwd(z, dy, P, bn, stats, act: int, G):
    """-> dz; writes G[bn.weight], G[bn.bias]."""
    N, C = z.shape[0], z.shape[1]
    Pn = z.numel() // (N * C)
    dz = torch.empty_like(z)
    ws = _bn_ws(C, z.device)
    check(lib().fdet_mbt_bn_bwd(ptr(z), ptr(dy), ptr(P[bn + ".weight"]), ptr(P[bn + ".bias"]), ptr(stats[0]), ptr(stats[1]), ptr(dz),
                                ptr(G[bn + ".weight"]), ptr(G[bn + ".bias"]), ptr(ws), ws.numel() * 4, N, C, Pn, act, stream()),
          "fdet_mbt_bn_bwd")
    return dz


def dw_fwd(x, w, k: int, s: int):
    N, C, H, W = x.shape
    Ho, Wo = (H, W) if s == 1 else ((H + 1) // 2, (W + 1) // 2)
    z = torch.empty(N, C, Ho, Wo, dtype=F32, device=x.device)
    check(lib().fdet_mbt_dw_fwd(ptr(x), ptr(w), ptr(z), N, C, H, W, k, s, stream()), "fdet_mbt_dw_fwd")
    return z


def dw_bwd(x, dz, w, k: int, s: int, dW):
    N, C, H, W = x.shape
    dx = torch.empty_like(x)
    ws = torch.empty(int(lib().fdet_mbt_taps_ws_bytes(C, k)) // 4 + 4, dtype=F32, device=x.device)
    check(lib().fdet_mbt_dw_bwd(ptr(x), ptr(dz), ptr(w), ptr(dx), ptr(dW), ptr(ws), ws.numel() * 4, N, C, H, W, k, s, stream()),
          "fdet_mbt_dw_bwd")
    return dx


def pw_fwd(x, w):
    """1x1 conv without bias: (N,Cin,H,W) -> (N,Cout,H,W); returns (z, (forward panel, backward panel))."""
    N, _, H, W = x.shape
    cout = w.shape[0]
    panels = hp.pointwise_pack(w)
    z = torch.empty(N, cout, H, W, dtype=F32, device=x.device)
    hp.pointwise_fwd(x, panels[0], None, z, slope=1.0)
    return z, panels


def pw_bwd(x, dz, panels, dW, add=None, need_dx: bool = True):
    hp.pointwise_wgrad(x, dz, dW, None)
    if not need_dx:
        return None
    dx = torch.empty_like(x)
    hp.pointwise_dgrad(dz, panels[1], dx, add=add)
    return dx


def se_fwd(x, P, p):
    N, C, H, W = x.shape
    R = P[p + ".conv_reduce.weight"].shape[0]
    dev = x.device
    pooled = torch.empty(N, C, dtype=F32, device=dev)
    hidden = torch.empty(N, R, dtype=F32, device=dev)
    pre = torch.empty(N, C, dtype=F32, device=dev)
    y = torch.empty_like(x)
    check(lib().fdet_mbt_se_fwd(ptr(x), ptr(P[p + ".conv_reduce.weight"]), ptr(P[p + ".conv_reduce.bias"]), ptr(P[p + ".conv_expand.weight"]),
                                ptr(P[p + ".conv_expand.bias"]), ptr(pooled), ptr(hidden), ptr(pre), ptr(y), N, C, R, H * W, stream()),
          "fdet_mbt_se_fwd")
    return y, (pooled, hidden, pre)


def se_bwd(x, dy, P, p, kept, G):
    N, C, H, W = x.shape
    R = P[p + ".conv_reduce.weight"].shape[0]
    dx = torch.empty_like(x)
    ws = torch.empty(2 * N * C + N * R + 4, dtype=F32, device=x.device)
    check(lib().fdet_mbt_se_bwd(ptr(x), ptr(dy), ptr(kept[0]), ptr(kept[1]), ptr(kept[2]), ptr(P[p + ".conv_reduce.weight"]),
                                ptr(P[p + ".conv_expand.weight"]), ptr(dx), ptr(G[p + ".conv_reduce.weight"]), ptr(G[p + ".conv_reduce.bias"]),
                                ptr(G[p + ".conv_expand.weight"]), ptr(G[p + ".conv_expand.bias"]), ptr(ws), ws.numel() * 4, N, C, R,
                                H * W, stream()), "fdet_mbt_se_bwd")
    return dx


class MobileNetTrainEngine:
    """forward_train(x, P) -> (y, saved); backward(saved, dy, P, G).  P: the module's state dict tensors on the GPU (fp32,
    contiguous; BatchNorm running statistics are updated IN PLACE), G: one gradient tensor per learnable parameter."""

    def forward_train(self, x: torch.Tensor, P: Dict[str, torch.Tensor]):
        if not x.is_cuda:
            raise hp.N.FdetError("the MobileNet training path runs on the GPU only (no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != 3 or x.dtype != F32:
            raise ValueError(f"MobileNet forward: expected (N,3,H,W) float32 in [0,1], got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        N, _, H, W = x.shape
        S: dict = {"x": x, "blocks": []}
        z = torch.empty(N, 16, (H + 1) // 2, (W + 1) // 2, dtype=F32, device=x.device)
        check(lib().fdet_mbt_stem_fwd(ptr(x), ptr(P["feature_extractor.0.weight"]), ptr(z), N, H, W, stream()), "fdet_mbt_stem_fwd")
        h, st = bn_fwd(z, P, "feature_extractor.1", ACT["hswish"])
        S["stem"] = (z, st)
        for bi, (kind, ci, ce, co, k, s, act, se) in enumerate(BLOCKS):
            p = block_prefix(bi)
            a = ACT[act]
            B: dict = {"in": h}
            if kind == "ir":
                z1, B["pw_panels"] = pw_fwd(h, P[p + ".conv_pw.weight"])
                e, B["bn1"] = bn_fwd(z1, P, p + ".bn1", a)
                B["z1"], B["e"] = z1, e
                dw_bn = ".bn2"
            else:
                e = h
                dw_bn = ".bn1"
            z2 = dw_fwd(e, P[p + ".conv_dw.weight"], k, s)
            d, B["bn_dw"] = bn_fwd(z2, P, p + dw_bn, a)
            B["z2"], B["d"] = z2, d
            if se:
                d2, B["se"] = se_fwd(d, P, p + ".se")
            else:
                d2 = d
            B["d2"] = d2
            last_w = ".conv_pwl.weight" if kind == "ir" else ".conv_pw.weight"
            last_bn = ".bn3" if kind == "ir" else ".bn2"
            z3, B["pwl_panels"] = pw_fwd(d2, P[p + last_w])
            skip = h if (s == 1 and ci == co) else None
            h, B["bn_last"] = bn_fwd(z3, P, p + last_bn, ACT["none"], residual=skip)
            B["z3"] = z3
            S["blocks"].append(B)
        zf, S["final_panels"] = pw_fwd(h, P["feature_extractor.3.5.0.conv.weight"])
        f, S["final_bn"] = bn_fwd(zf, P, "feature_extractor.3.5.0.bn1", ACT["hswish"])
        S["final_in"], S["zf"], S["f"] = h, zf, f
        Ns, _, Hs, Ws = f.shape
        y = torch.empty(Ns, 5, Hs, Ws, dtype=F32, device=f.device)
        hp.head_fwd(f, None, P["out.weight"], P["out.bias"], y, 3, 1)
        S["y"] = y
        return y, S

    def backward(self, S, dy: torch.Tensor, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor]) -> None:
        f, y = S["f"], S["y"]
        N = f.shape[0]
        dy = dy.to(F32).contiguous()
        ws = torch.empty(hp.head_bwd_ws_bytes(N, FEATURES, f.shape[2], f.shape[3], 3, 1) // 4 + 16, dtype=F32, device=f.device)
        df = torch.empty_like(f)
        hp.head_bwd(f, None, P["out.weight"], y, dy, df, G["out.weight"], G["out.bias"], ws, 3, 1)
        dz = bn_bwd(S["zf"], df, P, "feature_extractor.3.5.0.bn1", S["final_bn"], ACT["hswish"], G)
        dh = pw_bwd(S["final_in"], dz, S["final_panels"], G["feature_extractor.3.5.0.conv.weight"])
        for bi in reversed(range(len(BLOCKS))):
            kind, ci, ce, co, k, s, act, se = BLOCKS[bi]
            p = block_prefix(bi)
            a = ACT[act]
            B = S["blocks"][bi]
            has_skip = s == 1 and ci == co
            last_w = ".conv_pwl.weight" if kind == "ir" else ".conv_pw.weight"
            last_bn = ".bn3" if kind == "ir" else ".bn2"
            dz3 = bn_bwd(B["z3"], dh, P, p + last_bn, B["bn_last"], ACT["none"], G)     # (the skip branch's gradient is dh itself)
            dd2 = pw_bwd(B["d2"], dz3, B["pwl_panels"], G[p + last_w])
            dd = se_bwd(B["d"], dd2, P, p + ".se", B["se"], G) if se else dd2
            dw_bn = ".bn2" if kind == "ir" else ".bn1"
            dz2 = bn_bwd(B["z2"], dd, P, p + dw_bn, B["bn_dw"], a, G)
            e = B["e"] if kind == "ir" else B["in"]
            de = dw_bwd(e, dz2, P[p + ".conv_dw.weight"], k, s, G[p + ".conv_dw.weight"])
            if kind == "ir":
                dz1 = bn_bwd(B["z1"], de, P, p + ".bn1", B["bn1"], a, G)
                dh = pw_bwd(B["in"], dz1, B["pw_panels"], G[p + ".conv_pw.weight"], add=dh if has_skip else None)
            else:
                dh = de + dh if has_skip else de
        z, st = S["stem"]
        dzs = bn_bwd(z, dh, P, "feature_extractor.1", st, ACT["hswish"], G)
        x = S["x"]
        wst = torch.empty(int(lib().fdet_mbt_taps_ws_bytes(16, 0)) // 4 + 4, dtype=F32, device=x.device)
        check(lib().fdet_mbt_stem_wgrad(ptr(x), ptr(dzs), ptr(G["feature_extractor.0.weight"]), ptr(wst), wst.numel() * 4, x.shape[0],
                                        x.shape[2], x.shape[3], stream()), "fdet_mbt_stem_wgrad")


def learnable_names(state_names: List[str]) -> List[str]:
    return [n for n in state_names if not (n.endswith("running_mean") or n.endswith("running_var") or n.endswith("num_batches_tracked"))]


class MobileNetTrainFn(torch.autograd.Function):
    """Autograd bridge (as convstack.ConvStackFn): `loss.backward()` of a Lightning-style training_step reaches the
    hand-written backward; one gradient per learnable parameter, in `names` order."""

    @staticmethod
    def forward(ctx, engine: MobileNetTrainEngine, names, buffers, x, *params):
        P = {n: p.detach() for n, p in zip(names, params)}
        P.update(buffers)
        y, saved = engine.forward_train(x.detach(), P)
        ctx.engine, ctx.saved, ctx.names, ctx.P = engine, saved, names, P
        return y

    @staticmethod
    def backward(ctx, dy):
        G = {n: torch.empty_like(ctx.P[n]) for n in ctx.names}
        ctx.engine.backward(ctx.saved, dy, ctx.P, G)
        ctx.saved = None
        return (None, None, None, None) + tuple(G[n] for n in ctx.names)
