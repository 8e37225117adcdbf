"""Conv-stack engine: the PoolResnet / Resnet forward and hand-written backward as a fixed
sequence of HIP kernel launches (no autograd graph inside).

Reference: models/PoolResnet.py:11-105, models/Resnet.py:10-99 (forward); the backward is the
autograd of that forward, restated kernel by kernel:

    stem -> [ conv1+lrelu -> conv2+lrelu -> dropout2d*skip-add (-> maxpool) ] x blocks
         -> dropout2d(0.5) -> head conv -> sigmoid

Activations saved for backward per block: a (conv1 output), c (conv2 output, pre-dropout) and
the block input; everything else is recomputed in the fused tails.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import hotpath as hp
from . import ps as psm

F32 = torch.float32


@dataclass
class StackGeometry:
    kind: str
    filters: int
    in_ch: int
    H: int
    W: int
    S: int
    num_blocks: int
    stem_k: int
    stem_s: int
    stem_p: int
    head_k: int
    head_p: int
    pool_mult: int          # pool iff H > pool_mult*S  (PoolResnet.py:41: 2, Resnet.py:38: 1)

    def levels(self):
        """[(H_in, pool)] per block and the stem output size."""
        h0 = (self.H + 2 * self.stem_p - self.stem_k) // self.stem_s + 1
        w0 = (self.W + 2 * self.stem_p - self.stem_k) // self.stem_s + 1
        if h0 != w0:
            raise ValueError("square inputs only (reference swaps width/height, datasets/utils.py:107)")
        out, h = [], h0
        for _ in range(self.num_blocks):
            pool = 2 if h > self.pool_mult * self.S else 1
            if pool == 2 and h % 2:
                raise ValueError(f"cannot 2x2-pool an odd {h}x{h} map")
            out.append((h, pool))
            h //= pool
        s_out = h + 2 * self.head_p - self.head_k + 1
        if s_out != self.S:
            # the reference only fails later, in yolo_loss, with a shape mismatch
            raise ValueError(f"input {self.H}x{self.W} with this stem/head reaches a {s_out}x{s_out} grid, "
                             f"not num_of_patches={self.S}")
        return h0, out


PARAM_ORDER_DOC = "conv1.{weight,bias}, residual_blocks.k.conv{1,2}.{weight,bias}, out.{weight,bias}"


class KernelTimer:
    """HIP-event timing of kernel classes on the stream they are launched on (torch's current
    stream).  Events are only recorded while a timer is attached to the engine; nothing
    synchronises until summary()."""

    def __init__(self):
        self.rec = {}          # name -> [list of (start, end)], flops/launch, bytes/launch

    def span(self, name, flops=0.0, nbytes=0.0):
        return _Span(self, name, flops, nbytes)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, (evs, flops, nbytes) in self.rec.items():
            out[name] = (len(evs), sum(a.elapsed_time(b) for a, b in evs), flops / len(evs), nbytes / len(evs))
        return out


class _Span:
    def __init__(self, timer, name, flops, nbytes):
        self.t, self.name, self.flops, self.nbytes = timer, name, flops, nbytes

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        rec = self.t.rec.setdefault(self.name, [[], 0.0, 0.0])
        rec[0].append((self.a, self.b))
        rec[1] += self.flops
        rec[2] += self.nbytes


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOSPAN = _NoSpan()


class _PsScope(list):
    """PS buffers leased by one forward pass: [(pool key, PsTensor)]; they go back to the engine's pool when the scope
    is released explicitly or dropped (CPython: as soon as the saved state of the pass is)."""

    def __init__(self, pool):
        super().__init__()
        self.pool = pool

    def release(self):
        for key, t in self:
            self.pool.setdefault(key, []).append(t)
        self.clear()

    def __del__(self):
        self.release()


class ConvStack:
    """Owns packed weights, workspaces and saved activations; parameters are passed in as a
    dict of GPU tensors named like the reference's state_dict."""

    def __init__(self, geo: StackGeometry):
        self.geo = geo
        self.h0, self.lv = geo.levels()
        self._packed_key = None
        self._wpk: Dict[str, torch.Tensor] = {}
        self._ws: Dict[str, torch.Tensor] = {}
        self.slope = 0.2
        self.timer: Optional[KernelTimer] = None
        # conv arithmetic: "bf16x3" (bf16 hi/lo split on the bf16 matrix cores, fp32 accumulate,
        # ~1e-5 of fp32) where the channel count allows, else "f32" (exact fp32 MFMA chain).
        # FDET_PRECISION=f32 forces the exact path.
        import os
        want = os.environ.get("FDET_PRECISION", "bf16x3")
        self.x3 = (want in ("bf16x3", "bf16")) and hp.x3_supported(geo.filters, geo.filters)
        # pooled blocks: dropout*skip*maxpool (and its backward) inside the conv epilogues (FDET_POOL_FUSION=0: the
        # separate elementwise tail kernels of round 1)
        self.pool_fusion = self.x3 and os.environ.get("FDET_POOL_FUSION", "1") != "0"
        # weight gradients on a second HIP stream: they depend only on tensors the data-gradient chain has already
        # produced, so the batched weight-gradient launch of one resolution runs beside the data-gradient kernels of the
        # next (and the last one beside the stem's); joined before anything reads the gradients.  Opt-in
        # (FDET_WGRAD_STREAM=1): measured -0.6 % per step only -- both kernel families are bound by the same HBM/MFMA
        # resources, so they mostly slow each other down -- and overlapping launches make per-kernel timings (HIP events,
        # rocprofv3 --stats) meaningless, so the default keeps one stream
        self.wgrad_stream = os.environ.get("FDET_WGRAD_STREAM", "0") == "1"
        self._side = None
        # pre-split (PS) activations for the pooled 64-channel blocks (ps.py, csrc/fdet_ps.h): stem output, conv outputs
        # and their gradients live as bf16 hi|lo units, staged by LDS-DMA (FDET_PS=0: the fp32-I/O kernels of round 2)
        self.ps = self.pool_fusion and geo.filters == 64 and os.environ.get("FDET_PS", "1") != "0"
        self._ps_pool: Dict[tuple, list] = {}
        self.ps_strips = self.ps and os.environ.get("FDET_PS_STRIPS", "1") != "0"
        # precision16 (FDET_PRECISION=bf16 or set_precision("bf16")): the PS kernels run ONE bf16 MFMA pass on the hi planes
        # (bf16 activations and weights, fp32 accumulation / epilogues / master weights) -- the arithmetic of the reference's
        # Trainer(precision=16), train_model.py:50, with bf16 as the 16-bit type.  Only where the PS path runs (64 channels);
        # the stem, the head and the pooled-gradient routing keep their fp32-grade kernels.
        self.p16 = self.ps and want == "bf16"
        self._cur_N = 1                                    # batch size of the pass being planned (forward() sets it)
        # training head fused with the loss (fdet_head_loss_fused: forward + yolo_loss + their gradients in one kernel on
        # the matrix cores); FDET_HEAD_FUSED=0 keeps the separate head_fwd / yolo_loss / head_bwd launches
        self.head_fused = self.x3 and os.environ.get("FDET_HEAD_FUSED", "1") != "0"
        self._zero_ws: Dict[str, torch.Tensor] = {}

    def set_precision(self, name: str) -> None:
        """"bf16x3" (fp32-grade, default) or "bf16" (precision16: one MFMA pass; needs the PS path)."""
        if name not in ("bf16x3", "bf16"):
            raise ValueError("precision must be 'bf16x3' or 'bf16'")
        if name == "bf16" and not self.ps:
            raise ValueError("precision16 needs the pre-split path (64 channels, bf16x3-capable geometry)")
        self.p16 = name == "bf16"
        self._ps_pool.clear()                              # a bf16x3 pass may have left lo planes in pooled buffers

    def u8_frames_ok(self) -> bool:
        """forward(..., u8_frames=True) is available: the PoolResnet stem writes the first block's PS input itself, so it can
        read the uint8 frames directly (x / 255 fused into its staging)."""
        g = self.geo
        return bool(self.ps and self.x3 and self.lv and 30 <= self.h0 <= 62 and self.lv[0][1] == 2 and g.W % 4 == 0 and
                    hp.stem_x3_supported(g.in_ch, g.W, g.stem_k, g.stem_s, g.stem_p) and g.filters == 64)

    def head_loss_fusable(self) -> bool:
        """The fused training head (forward + loss + backward of the head in one launch) covers this geometry."""
        g = self.geo
        hl = self.lv[-1][0] // self.lv[-1][1] if self.lv else self.h0
        return self.head_fused and hp.head_loss_fused_supported(g.filters, hl, hl, g.head_k, g.head_p)

    def _t(self, kind: str, N: int, h: int, flops: float = 0.0, nbytes: float = 0.0):
        if self.timer is None:
            return _NOSPAN
        return self.timer.span(f"{kind}@{h}x{h}", flops, nbytes)

    def _conv_flops(self, N: int, h: int) -> float:
        F_ = self.geo.filters
        return 2.0 * N * F_ * F_ * 9 * h * h

    def _act_bytes(self, N: int, h: int, tensors: float) -> float:
        """Algorithmic HBM bytes of `tensors` activation-sized fp32 tensors (SURVEY.md 8d:
        every input read once, every output written once; weights are L2-resident)."""
        return 4.0 * N * self.geo.filters * h * h * tensors

    # ------------------------------------------------------------------ weights
    def _ensure_packed(self, P: Dict[str, torch.Tensor], force: bool = False):
        key = (self.x3,) + tuple((P[k].data_ptr(), P[k]._version) for k in sorted(P) if k.endswith("weight"))
        if not force and key == self._packed_key:
            return
        F_ = self.geo.filters
        nf, nb = hp.packed_sizes(F_, F_)
        dev = P["conv1.weight"].device
        names = [f"residual_blocks.{k}.conv{j}" for k in range(self.geo.num_blocks) for j in (1, 2)]
        for name in names:
            if name + ".f" not in self._wpk:
                self._wpk[name + ".f"] = torch.empty(nf, dtype=F32, device=dev)
                self._wpk[name + ".b"] = torch.empty(nb, dtype=F32, device=dev)
        if self.x3:                                        # every layer in one launch
            hp.pack_conv3x3_weights_batched([P[nm + ".weight"] for nm in names], [self._wpk[nm + ".f"] for nm in names],
                                            [self._wpk[nm + ".b"] for nm in names])
        else:
            for name in names:
                hp.pack_conv3x3_weights(P[name + ".weight"], self._wpk[name + ".f"], self._wpk[name + ".b"], x3=False)
        self._packed_key = key

    def mark_params_dirty(self):
        """Call after the parameters were updated behind torch's back (fdet_adam_step)."""
        self._packed_key = None

    def _workspace(self, name: str, nbytes: int, dev) -> torch.Tensor:
        n = (nbytes + 3) // 4
        t = self._ws.get(name)
        if t is None or t.numel() < n or t.device != dev:
            t = torch.empty(max(n, 4), dtype=F32, device=dev)
            self._ws[name] = t
        return t

    def _fused_pool(self, hk: int, N: int = 1) -> bool:
        return self.pool_fusion and hp.pool_fusion_supported(self.geo.filters, self.geo.filters, hk, hk, N)

    # ------------------------------------------------------------------ PS buffers
    def _ps_block(self, k: int) -> bool:
        """Block k runs on pre-split activations: a pooled block whose even map is 30..62 columns wide, or wider and kept
        as column strips (csrc/fdet_ps.h; FDET_PS_STRIPS=0: the round-2 kernels for wide maps)."""
        if not self.ps or k >= len(self.lv):
            return False
        hk, pool = self.lv[k]
        if pool != 2 or hk % 2 or hk < 30:
            return False
        # (the batch-dependent limits of run_ps: 32-bit byte offsets of the fp32 tensors the pooled modes touch)
        if self._cur_N * self.geo.filters * hk * hk * 4 >= 2 ** 31:
            return False
        if hk > 62:
            return self.ps_strips and psm.strips_of(hk)[0] > 1 and psm.conv3x3_wgrad_ps_ws_bytes(1, self._cur_N, self.geo.filters, hk, hk) > 0
        return self._fused_pool(hk, self._cur_N)

    def _strips(self, hk: int) -> bool:
        return hk > 62

    def _ps_chain(self, k: int) -> bool:
        """The chain run starting at block k keeps its per-block tensors in PS (fdet_block_chain_*_ps) and its weight
        gradients come from the PS kernel."""
        if not self.ps or k >= len(self.lv) or self._chain_run(k) <= 1:
            return False
        hk = self.lv[k][0]
        return psm.conv3x3_wgrad_ps_ws_bytes(1, self._cur_N, self.geo.filters, hk, hk) > 0 and \
            self._cur_N * self.geo.filters * hk * hk * 4 < 2 ** 31

    def _ps_take(self, scope: "_PsScope", N: int, C: int, H: int, W: int, dev) -> "psm.PsTensor":
        """A zero-haloed PS buffer from the engine's pool; it returns to the pool when `scope` (kept alive by the saved
        state of a forward pass, or released at the end of an inference call) goes away.  Producers write real elements
        only, so a recycled buffer still has zero halos."""
        key = (N, C, H, W, str(dev))
        free = self._ps_pool.setdefault(key, [])
        t = free.pop() if free else psm.PsTensor(N, C, H, W, dev)
        scope.append((key, t))
        return t

    def _chain_run(self, k: int) -> int:
        """Number of consecutive un-pooled blocks starting at block k that can run as one
        LDS-resident chain (bf16x3, 64 channels, small maps); 0/1 = use the per-layer kernels."""
        import os
        if not self.x3 or os.environ.get("FDET_CHAIN", "1") == "0":
            return 0
        hk, pool = self.lv[k]
        if pool != 1 or not hp.block_chain_supported(self.geo.filters, hk, hk):
            return 0
        run = 1
        while k + run < len(self.lv) and self.lv[k + run] == (hk, 1) and run < 16:
            run += 1
        return run

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, P: Dict[str, torch.Tensor], masks: Optional[Dict[str, torch.Tensor]] = None,
                save: bool = False, loss_targets: Optional[torch.Tensor] = None, G: Optional[Dict[str, torch.Tensor]] = None,
                u8_frames: bool = False):
        """x (N,C,H,W) f32 on the GPU -> y (N,5,S,S).  masks: per-(n,c) dropout scales
        {"residual_blocks.k": (N,F), "head": (N,F)} or None (eval).  save=True keeps what
        backward needs and returns it as the second value.
        loss_targets (N,5,S,S) + G (gradient views): the head runs FUSED with yolo_loss and its own backward
        (head_loss_fusable()): saved["loss"] = (loss_per_image, loss_sum), G["out.*"] are written, and backward() is
        called with dy=None."""
        g = self.geo
        if x.dim() != 4 or tuple(x.shape[1:]) != (g.in_ch, g.H, g.W):
            raise ValueError(f"expected input (N,{g.in_ch},{g.H},{g.W}), got {tuple(x.shape)}")
        if u8_frames:
            # inference on the uint8 frames themselves: x / 255 happens inside the stem (fdet_stem_fwd_ps_u8)
            if x.dtype != torch.uint8 or save or not self.u8_frames_ok():
                raise ValueError("u8_frames: uint8 input, inference only, PoolResnet stem on the pre-split path")
            x = x.contiguous()
        elif x.dtype != F32 or not x.is_contiguous():
            x = x.to(F32).contiguous()
        self._ensure_packed(P)
        N, F_, dev = x.shape[0], g.filters, x.device
        if N != self._cur_N:
            # PS buffers are sized per batch: a pool filled for another N (a last partial batch, a validation batch) would
            # only pile up ~260 MB sets outside the caching allocator -- drop it
            self._ps_pool.clear()
            self._cur_N = N
        ws = self._workspace("stem", hp.stem_ws_bytes(N, g.in_ch, F_, g.H, g.W, g.stem_k, g.stem_s, g.stem_p), dev)
        # the stem writes the first block's PS input itself: the PoolResnet stem (k10 s8 p2, plain layout) or the Resnet stem
        # (k3 s2 p1, column strips included)
        stem_ps = self._ps_block(0) and self.x3 and \
            ((not self._strips(self.h0) and hp.stem_x3_supported(g.in_ch, g.W, g.stem_k, g.stem_s, g.stem_p)) or
             hp.stem_k3_fwd_ps_supported(g.in_ch, F_, g.H, g.W, g.stem_k, g.stem_s, g.stem_p))
        if u8_frames and not stem_ps:                      # (a batch too large for the pre-split path: the separate x / 255)
            x = hp.u8_to_f32_norm(x)
        h = torch.empty(N, F_, self.h0, self.h0, dtype=F32, device=dev) if not stem_ps else None
        stem_flops = 2.0 * N * F_ * g.in_ch * g.stem_k * g.stem_k * self.h0 * self.h0
        stem_bytes = 4.0 * N * (g.in_ch * g.H * g.W + F_ * self.h0 * self.h0)
        scope = _PsScope(self._ps_pool)
        h_ps = None
        stem_x3 = self.x3 and hp.stem_x3_supported(g.in_ch, g.W, g.stem_k, g.stem_s, g.stem_p)
        with self._t("stem_fwd", N, self.h0, stem_flops, stem_bytes):
            if stem_ps:
                h, h_ps = None, self._ps_take(scope, N, F_, self.h0, self.h0, dev)
                psm.stem_fwd_ps(x, P["conv1.weight"], P["conv1.bias"], h_ps, g.stem_k, g.stem_s, g.stem_p, p16=self.p16)
            else:
                hp.stem_fwd(x, P["conv1.weight"], P["conv1.bias"], h, ws, g.stem_k, g.stem_s, g.stem_p, x3=stem_x3)
        saved = {"x": x, "blocks": [], "masks": masks, "ps_scope": scope} if save else None
        k = -1
        while k + 1 < len(self.lv):
            k += 1
            hk, pool = self.lv[k]
            if self._ps_block(k):
                # pooled block on pre-split activations: conv1 -> a (PS); conv2 + tail -> pooled output (PS when the next
                # block is a PS block too, else fp32 NCHW) + channel-innermost routing bytes
                name = f"residual_blocks.{k}"
                sc = masks[name] if masks is not None else None
                if h_ps is None:
                    h_ps = psm.PsTensor.from_f32(h, out=self._ps_take(scope, N, F_, hk, hk, dev))
                a_ps = self._ps_take(scope, N, F_, hk, hk, dev)
                with self._t("conv3x3_fwd", N, hk, self._conv_flops(N, hk), self._act_bytes(N, hk, 2)):
                    psm.conv3x3_ps_fwd(h_ps, self._wpk[name + ".conv1.f"], P[name + ".conv1.bias"], a_ps, self.slope, p16=self.p16)
                    psm.halo_exchange(a_ps, p16=self.p16)      # strips: conv2 (and the weight gradient) read a's edge columns
                # (a strip level hands its pooled output on as fp32 NCHW; the next level's converter writes its halos)
                nxt = (self._ps_block(k + 1) or self._ps_chain(k + 1)) and not self._strips(hk)
                out_ps = self._ps_take(scope, N, F_, hk // 2, hk // 2, dev) if nxt else None
                out = None if nxt else torch.empty(N, F_, hk // 2, hk // 2, dtype=F32, device=dev)
                route = psm.route8_like(N, F_, hk, hk, dev) if save else None
                with self._t("conv3x3_fwd_pool", N, hk, self._conv_flops(N, hk), self._act_bytes(N, hk, 2 + 0.25 + (1 / 16 if save else 0))):
                    psm.conv3x3_ps_fwd_pool(a_ps, self._wpk[name + ".conv2.f"], P[name + ".conv2.bias"], h_ps, sc, out_ps, out,
                                            route, self.slope, p16=self.p16)
                if save:
                    saved["blocks"].append((h_ps, a_ps, route))
                h, h_ps = out, out_ps
                continue
            run = self._chain_run(k)
            if run > 1 and self._ps_chain(k):
                # blocks k .. k+run-1 keep their activation on the CU: one launch (fdet_block_chain_fwd_ps); what backward
                # needs is kept in PS (a_k, block outputs: operands of the weight gradients; c_k: hi plane = its signs)
                names = [f"residual_blocks.{kk}" for kk in range(k, k + run)]
                if h_ps is None:
                    h_ps = psm.PsTensor.from_f32(h, out=self._ps_take(scope, N, F_, hk, hk, dev))
                a_l = [self._ps_take(scope, N, F_, hk, hk, dev) for _ in names] if save else None
                c_l = [self._ps_take(scope, N, F_, hk, hk, dev) for _ in names] if save else None
                o_l = [self._ps_take(scope, N, F_, hk, hk, dev) for _ in names[:-1]] if save else None
                out = torch.empty(N, F_, hk, hk, dtype=F32, device=dev)
                with self._t("chain_fwd", N, hk, 2 * run * self._conv_flops(N, hk), self._act_bytes(N, hk, 1 + (2.5 * run if save else 1))):
                    psm.block_chain_fwd_ps(h_ps, [self._wpk[nm + ".conv1.f"] for nm in names], [P[nm + ".conv1.bias"] for nm in names],
                                           [self._wpk[nm + ".conv2.f"] for nm in names], [P[nm + ".conv2.bias"] for nm in names],
                                           [masks[nm] for nm in names] if masks is not None else None, a_l, c_l, o_l, out, self.slope,
                                           p16=self.p16)
                if save:
                    for i in range(run):
                        saved["blocks"].append((h_ps if i == 0 else o_l[i - 1], a_l[i], c_l[i]))
                h, h_ps = out, None
                k += run - 1
                continue
            if run > 1:
                # the same with fp32 NCHW tensors (fdet_block_chain_fwd_bf16x3)
                names = [f"residual_blocks.{kk}" for kk in range(k, k + run)]
                a_l = [torch.empty(N, F_, hk, hk, dtype=F32, device=dev) for _ in names] if save else None
                c_l = [torch.empty(N, F_, hk, hk, dtype=F32, device=dev) for _ in names] if save else None
                o_l = [torch.empty(N, F_, hk, hk, dtype=F32, device=dev) if (save or i == run - 1) else None
                       for i in range(run)]
                with self._t("chain_fwd", N, hk, 2 * run * self._conv_flops(N, hk), self._act_bytes(N, hk, 1 + (3 * run if save else 1))):
                    hp.block_chain_fwd(h, [self._wpk[nm + ".conv1.f"] for nm in names], [P[nm + ".conv1.bias"] for nm in names],
                                       [self._wpk[nm + ".conv2.f"] for nm in names], [P[nm + ".conv2.bias"] for nm in names],
                                       [masks[nm] for nm in names] if masks is not None else None, a_l, c_l, o_l, self.slope)
                if save:
                    for i in range(run):
                        saved["blocks"].append((h if i == 0 else o_l[i - 1], a_l[i], c_l[i]))
                h = o_l[-1]
                k += run - 1
                continue
            name = f"residual_blocks.{k}"
            sc = masks[name] if masks is not None else None
            a = torch.empty(N, F_, hk, hk, dtype=F32, device=dev)
            with self._t("conv3x3_fwd", N, hk, self._conv_flops(N, hk), self._act_bytes(N, hk, 2)):
                hp.conv3x3_fwd(h, self._wpk[name + ".conv1.f"], P[name + ".conv1.bias"], F_, y_full=a, slope=self.slope, x3=self.x3)
            out = torch.empty(N, F_, hk // pool, hk // pool, dtype=F32, device=dev)
            if pool == 2 and self._fused_pool(hk, N):
                # conv2 + lrelu + dropout*skip + maxpool in one kernel; c is never written, backward gets one
                # routing byte per pooling window instead
                c = torch.empty(N, F_, hk // 2, hk // 2, dtype=torch.uint8, device=dev) if save else None
                with self._t("conv3x3_fwd_pool", N, hk, self._conv_flops(N, hk), self._act_bytes(N, hk, 2 + 0.25 + (1 / 16 if save else 0))):
                    hp.conv3x3_fwd_pool(a, self._wpk[name + ".conv2.f"], P[name + ".conv2.bias"], h, sc, out, c, self.slope)
            elif pool == 2:
                c = torch.empty_like(a)
                with self._t("conv3x3_fwd", N, hk, self._conv_flops(N, hk), self._act_bytes(N, hk, 2)):
                    hp.conv3x3_fwd(a, self._wpk[name + ".conv2.f"], P[name + ".conv2.bias"], F_, y_full=c, slope=self.slope, x3=self.x3)
                with self._t("tail_fwd", N, hk, 0.0, self._act_bytes(N, hk, 2.25)):
                    hp.block_tail_fwd(c, h, sc, out, 2)
            else:
                c = torch.empty_like(a) if save else None
                with self._t("conv3x3_fwd", N, hk, self._conv_flops(N, hk), self._act_bytes(N, hk, 4 if save else 3)):
                    hp.conv3x3_fwd(a, self._wpk[name + ".conv2.f"], P[name + ".conv2.bias"], F_, y_full=c, skip=h,
                                   drop_scale=sc, y_out=out, slope=self.slope, x3=self.x3)
            if save:
                saved["blocks"].append((h, a, c))
            h = out
        y = torch.empty(N, 5, g.S, g.S, dtype=F32, device=dev)
        if loss_targets is not None and save and G is not None and self.head_loss_fusable():
            if tuple(loss_targets.shape) != tuple(y.shape):
                raise ValueError(f"loss targets {tuple(loss_targets.shape)} != head output {tuple(y.shape)}")
            tg = loss_targets if (loss_targets.dtype == F32 and loss_targets.is_contiguous()) else loss_targets.to(F32).contiguous()
            hl = h.shape[2]
            nb = hp.head_loss_fused_ws_bytes(N, F_, hl, hl, g.head_k, g.head_p)
            ws = self._zero_ws.get("head_fused")
            if ws is None or ws.numel() * 4 != ((nb + 3) // 4) * 4 or ws.device != dev:
                ws = self._zero_ws["head_fused"] = torch.zeros((nb + 3) // 4, dtype=F32, device=dev)   # ticket counter starts at zero
            lpi = torch.empty(N, dtype=F32, device=dev)
            lsum = torch.empty(1, dtype=F32, device=dev)
            dout = torch.empty_like(h)
            macs = N * F_ * g.head_k ** 2 * (2 * 5 * g.S ** 2 + 5 * hl * hl)          # forward + dW + dx
            with self._t("head_loss_fused", N, hl, 2.0 * macs, self._act_bytes(N, hl, 2)):
                hp.head_loss_fused(h, masks["head"] if masks is not None else None, P["out.weight"], P["out.bias"], tg, y, lpi, lsum,
                                   dout, G["out.weight"], G["out.bias"], ws, g.head_k, g.head_p)
            saved["loss"] = (lpi, lsum)
            saved["head_dout"] = dout
        else:
            with self._t("head_fwd", N, h.shape[2], 2.0 * N * 5 * F_ * g.head_k ** 2 * g.S ** 2, self._act_bytes(N, h.shape[2], 1)):
                hp.head_fwd(h, masks["head"] if masks is not None else None, P["out.weight"], P["out.bias"], y, g.head_k, g.head_p)
        if save:
            saved["h_last"] = h
            saved["y"] = y
        else:
            scope.release()                                # inference: every PS buffer of the pass is free again
        return y, saved

    # ------------------------------------------------------------------ backward
    def backward(self, saved, dy: torch.Tensor, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor],
                 after_block=None) -> None:
        """dy = d loss / d y (N,5,S,S).  Writes every parameter gradient into G[name]
        (overwrites; same names/shapes as P).  `after_block(k)` is called once block k's
        gradients have been enqueued (data-parallel bucket launch)."""
        g = self.geo
        F_ = g.filters
        x, masks = saved["x"], saved["masks"]
        N, dev = x.shape[0], x.device
        self._cur_N = N                                    # the path decisions below are those of this pass's forward
        h_last, y = saved["h_last"], saved["y"]
        if saved.get("head_dout") is not None:
            dout = saved["head_dout"]                      # the fused head already ran its backward (G["out.*"] written)
        else:
            if dy is None or tuple(dy.shape) != tuple(y.shape):
                raise ValueError(f"dy shape {None if dy is None else tuple(dy.shape)} != y shape {tuple(y.shape)}")
            dy = dy.to(F32).contiguous()
            hl = h_last.shape[2]
            ws = self._workspace("head", hp.head_bwd_ws_bytes(N, F_, hl, hl, g.head_k, g.head_p), dev)
            dout = torch.empty_like(h_last)
            with self._t("head_bwd", N, hl, 4.0 * N * 5 * F_ * g.head_k ** 2 * g.S ** 2, self._act_bytes(N, hl, 2)):
                hp.head_bwd(h_last, masks["head"] if masks is not None else None, P["out.weight"], y, dy, dout,
                            G["out.weight"], G["out.bias"], ws, g.head_k, g.head_p)
        pending = []          # (x, dz, weight name) of same-resolution convs awaiting one batched wgrad launch

        side_keep = []        # tensors in use on the side stream: referenced until the streams are joined
        main = torch.cuda.current_stream(dev)
        side = None
        if self.wgrad_stream:
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side

        def join():
            """Everything queued on the side stream happens before whatever the main stream is given next."""
            if side is not None and side_keep:
                main.wait_stream(side)
                side_keep.clear()

        def flush(hk_):
            if not pending:
                return
            fl_ = self._conv_flops(N, hk_)
            if side is not None:
                side.wait_stream(main)                    # the operands were produced on the main stream
            with (torch.cuda.stream(side) if side is not None else _NOSPAN):
                for i0 in range(0, len(pending), 16):
                    grp = pending[i0:i0 + 16]
                    wsb_ = self._workspace("wgrad_batched", hp.conv3x3_wgrad_batched_ws_bytes(len(grp), N, F_, F_, hk_, hk_), dev)
                    with self._t("conv3x3_wgrad", N, hk_, fl_ * len(grp), self._act_bytes(N, hk_, 2) * len(grp)):
                        hp.conv3x3_wgrad_batched([p_[0] for p_ in grp], [p_[1] for p_ in grp],
                                                 [G[p_[2] + ".weight"] for p_ in grp], [G[p_[2] + ".bias"] for p_ in grp], wsb_)
            if side is not None:
                side_keep.extend(pending)
            pending.clear()

        pending_ps = []       # the same for blocks on pre-split activations: (x PS, dz PS, weight name)

        def flush_ps(hk_):
            if not pending_ps:
                return
            fl_ = self._conv_flops(N, hk_)
            for i0 in range(0, len(pending_ps), 16):
                grp = pending_ps[i0:i0 + 16]
                wsb_ = self._workspace("wgrad_ps", psm.conv3x3_wgrad_ps_ws_bytes(len(grp), N, F_, hk_, hk_), dev)
                with self._t("conv3x3_wgrad", N, hk_, fl_ * len(grp), self._act_bytes(N, hk_, 2) * len(grp)):
                    psm.conv3x3_wgrad_ps_batched([p_[0] for p_ in grp], [p_[1] for p_ in grp],
                                                 [G[p_[2] + ".weight"] for p_ in grp], [G[p_[2] + ".bias"] for p_ in grp], wsb_,
                                                 p16=self.p16)
            pending_ps.clear()

        # runs of blocks that went through the forward chain come back through the backward chain
        chain_start = {}
        kk = 0
        while kk < g.num_blocks:
            run = self._chain_run(kk)
            if run > 1:
                chain_start[kk + run - 1] = kk
                kk += run
            else:
                kk += 1
        k = g.num_blocks
        while k > 0:
            k -= 1
            hk, pool = self.lv[k]
            if k in chain_start and self._ps_chain(chain_start[k]):
                k0 = chain_start[k]
                ks = list(range(k0, k + 1))
                names = [f"residual_blocks.{q}" for q in ks]
                scope = saved.get("ps_scope")
                if scope is None:
                    scope = saved["ps_scope"] = _PsScope(self._ps_pool)

                def as_ps(t):                             # (a caller may have replaced saved tensors by fp32 NCHW ones)
                    if isinstance(t, psm.PsTensor):
                        return t
                    return psm.PsTensor.from_f32(t.to(F32).contiguous(), out=self._ps_take(scope, N, F_, hk, hk, dev))
                x_l = [as_ps(saved["blocks"][q][0]) for q in ks]
                a_l = [as_ps(saved["blocks"][q][1]) for q in ks]
                c_l = [as_ps(saved["blocks"][q][2]) for q in ks]
                dz1_l = [self._ps_take(scope, N, F_, hk, hk, dev) for _ in ks]
                dz2_l = [self._ps_take(scope, N, F_, hk, hk, dev) for _ in ks]
                dx = torch.empty_like(dout)
                with self._t("chain_bwd", N, hk, 2 * len(ks) * self._conv_flops(N, hk), self._act_bytes(N, hk, 2 + 3 * len(ks))):
                    psm.block_chain_bwd_ps(dout, [self._wpk[nm + ".conv1.b"] for nm in names], [self._wpk[nm + ".conv2.b"] for nm in names],
                                           [masks[nm] for nm in names] if masks is not None else None, a_l, c_l, dz1_l, dz2_l, dx, self.slope,
                                           p16=self.p16)
                for i in reversed(range(len(ks))):
                    pending_ps.append((a_l[i], dz2_l[i], names[i] + ".conv2"))
                    pending_ps.append((x_l[i], dz1_l[i], names[i] + ".conv1"))
                dout = dx
                k = k0
                if k == 0 or self.lv[k - 1][0] != hk:
                    flush_ps(hk)
                    if after_block is not None:
                        for q in range(k, g.num_blocks):
                            if self.lv[q][0] == hk:
                                after_block(q)
                continue
            if k in chain_start:
                k0 = chain_start[k]
                ks = list(range(k0, k + 1))
                names = [f"residual_blocks.{q}" for q in ks]
                dz1_l = [torch.empty_like(saved["blocks"][q][1]) for q in ks]
                dz2_l = [torch.empty_like(saved["blocks"][q][1]) for q in ks]
                dx = torch.empty_like(dout)
                with self._t("chain_bwd", N, hk, 2 * len(ks) * self._conv_flops(N, hk), self._act_bytes(N, hk, 2 + 4 * len(ks))):
                    hp.block_chain_bwd(dout, [self._wpk[nm + ".conv1.b"] for nm in names], [self._wpk[nm + ".conv2.b"] for nm in names],
                                       [masks[nm] for nm in names] if masks is not None else None,
                                       [saved["blocks"][q][1] for q in ks], [saved["blocks"][q][2] for q in ks],
                                       dz1_l, dz2_l, dx, self.slope)
                for i in reversed(range(len(ks))):
                    q = ks[i]
                    pending.append((saved["blocks"][q][1], dz2_l[i], names[i] + ".conv2"))
                    pending.append((saved["blocks"][q][0], dz1_l[i], names[i] + ".conv1"))
                dout = dx
                k = k0
                if k == 0 or self.lv[k - 1][0] != hk:
                    flush(hk)
                    if after_block is not None:
                        join()                            # the gradient bucket is about to be reduced
                        for q in range(k, g.num_blocks):
                            if self.lv[q][0] == hk:
                                after_block(q)
                continue
            name = f"residual_blocks.{k}"
            xin, a, c = saved["blocks"][k]
            sc = masks[name] if masks is not None else None
            if self._ps_block(k) and c is not None and c.dtype == torch.uint8 and \
                    (isinstance(a, psm.PsTensor) or isinstance(xin, psm.PsTensor) or c.dim() == 5):
                # pooled block on pre-split activations (a caller may have replaced saved tensors by fp32 / NCHW ones)
                scope = saved.get("ps_scope")
                if scope is None:
                    scope = saved["ps_scope"] = _PsScope(self._ps_pool)
                if not isinstance(xin, psm.PsTensor):
                    xin = psm.PsTensor.from_f32(xin.to(F32).contiguous(), out=self._ps_take(scope, N, F_, hk, hk, dev))
                if not isinstance(a, psm.PsTensor):
                    a = psm.PsTensor.from_f32(a.to(F32).contiguous(), out=self._ps_take(scope, N, F_, hk, hk, dev))
                strips = self._strips(hk)
                if c.dim() == 4:                           # routing bytes in NCHW -> channel-innermost
                    if strips:
                        raise ValueError("routing bytes of a strip level are per strip (ps.route8_shape)")
                    c = c.view(N, F_ // 8, 8, hk // 2, hk // 2).permute(0, 1, 3, 4, 2).contiguous()
                fl = self._conv_flops(N, hk)

                def wgrad_now(x_, dz_, nm_, clean=False):
                    # strips: the weight gradient wants ZERO halo slots in dz (a halo is not a position of its strip), the
                    # data-gradient conv behind it the neighbour columns -- so each layer's weight gradient runs right
                    # here, between the two halo passes, instead of in the level's batched launch (clean: the producer
                    # cleared the halo slots itself)
                    if not clean:
                        psm.halo_exchange(dz_, zero_only=True, p16=self.p16)
                    pending_ps.append((x_, dz_, nm_))
                    flush_ps(hk)
                    psm.halo_exchange(dz_, p16=self.p16)
                dz2 = self._ps_take(scope, N, F_, hk, hk, dev)
                with self._t("pool_route_bwd", N, hk, 0.0, self._act_bytes(N, hk, 1 + 0.25 + 1 / 16)):
                    psm.pool_route_bwd_ps(dout, c, sc, dz2, self.slope, p16=self.p16)
                if strips:
                    wgrad_now(a, dz2, name + ".conv2", clean=True)      # fdet_pool_route_bwd_ps clears dz2's halo slots
                dz1 = self._ps_take(scope, N, F_, hk, hk, dev)
                with self._t("conv3x3_dgrad", N, hk, fl, self._act_bytes(N, hk, 3)):
                    psm.conv3x3_ps_dgrad_act(dz2, self._wpk[name + ".conv2.b"], a, dz1, self.slope, p16=self.p16)
                if strips:
                    wgrad_now(xin, dz1, name + ".conv1")
                dx = torch.empty(N, F_, hk, hk, dtype=F32, device=dev)
                with self._t("conv3x3_dgrad_unpool", N, hk, fl, self._act_bytes(N, hk, 2 + 0.25 + 1 / 16)):
                    psm.conv3x3_ps_dgrad_unpool(dz1, self._wpk[name + ".conv1.b"], dout, c, dx, self.slope, p16=self.p16)
                if not strips:
                    pending_ps.append((a, dz2, name + ".conv2"))
                    pending_ps.append((xin, dz1, name + ".conv1"))
                dout = dx
                if k == 0 or self.lv[k - 1][0] != hk or not self._ps_block(k - 1):
                    flush_ps(hk)
                    if after_block is not None:
                        for kk in range(k, g.num_blocks):
                            if self.lv[kk][0] == hk:
                                after_block(kk)
                continue
            dz2 = torch.empty_like(a)
            fused_pool = pool == 2 and c.dtype == torch.uint8      # forward kept routing bytes instead of c
            de = torch.empty_like(a) if (pool == 2 and not fused_pool) else None
            if fused_pool:
                with self._t("pool_route_bwd", N, hk, 0.0, self._act_bytes(N, hk, 1 + 0.25 + 1 / 16)):
                    hp.pool_route_bwd(dout, c, sc, dz2, self.slope)
            else:
                with self._t("tail_bwd", N, hk, 0.0, self._act_bytes(N, hk, 4.25 if pool == 2 else 3)):
                    hp.block_tail_bwd(dout, c, xin, sc, dz2, de, pool, self.slope)
            if pool == 1:
                de = dout
            fl = self._conv_flops(N, hk)
            batched = self.x3 and hp.wgrad_x3_supported(N, F_, F_, hk, hk)
            if not batched:
                wws = self._workspace("wgrad", hp.conv3x3_wgrad_ws_bytes(N, F_, F_, hk, hk), dev)
                with self._t("conv3x3_wgrad", N, hk, fl, self._act_bytes(N, hk, 2)):
                    hp.conv3x3_wgrad(a, dz2, G[name + ".conv2.weight"], G[name + ".conv2.bias"], wws)
            dz1 = torch.empty_like(a)
            with self._t("conv3x3_dgrad", N, hk, fl, self._act_bytes(N, hk, 3)):
                hp.conv3x3_dgrad(dz2, self._wpk[name + ".conv2.b"], F_, dz1, act=a, slope=self.slope, x3=self.x3)
            if not batched:
                with self._t("conv3x3_wgrad", N, hk, fl, self._act_bytes(N, hk, 2)):
                    hp.conv3x3_wgrad(xin, dz1, G[name + ".conv1.weight"], G[name + ".conv1.bias"], wws)
            dx = torch.empty_like(a) if batched else dz2      # batched: dz2 stays alive until the flush
            if fused_pool:
                with self._t("conv3x3_dgrad_unpool", N, hk, fl, self._act_bytes(N, hk, 2 + 0.25 + 1 / 16)):
                    hp.conv3x3_dgrad_unpool(dz1, self._wpk[name + ".conv1.b"], F_, dout, c, dx, self.slope)
            else:
                with self._t("conv3x3_dgrad", N, hk, fl, self._act_bytes(N, hk, 3)):
                    hp.conv3x3_dgrad(dz1, self._wpk[name + ".conv1.b"], F_, dx, add=de, slope=self.slope, x3=self.x3)
            dout = dx
            if batched:
                pending.append((a, dz2, name + ".conv2"))
                pending.append((xin, dz1, name + ".conv1"))
                # weight gradients of a run of same-resolution blocks go out in one launch
                if k == 0 or self.lv[k - 1][0] != hk:
                    flush(hk)
                    if after_block is not None:
                        join()
                        for kk in range(k, g.num_blocks):
                            if self.lv[kk][0] == hk:
                                after_block(kk)
            elif after_block is not None:
                after_block(k)
        ws = self._workspace("stem", hp.stem_ws_bytes(N, g.in_ch, F_, g.H, g.W, g.stem_k, g.stem_s, g.stem_p), dev)
        stem_flops = 2.0 * N * F_ * g.in_ch * g.stem_k * g.stem_k * self.h0 * self.h0
        with self._t("stem_wgrad", N, self.h0, stem_flops, 4.0 * N * (g.in_ch * g.H * g.W + F_ * self.h0 * self.h0)):
            stem_x3 = self.x3 and ((g.W % 16 == 0 and hp.stem_x3_supported(g.in_ch, g.W, g.stem_k, g.stem_s, g.stem_p)) or
                                   hp.stem_k3_wgrad_x3_supported(g.in_ch, F_, g.H, g.W, g.stem_k, g.stem_s, g.stem_p))
            hp.stem_wgrad(x, dout, G["conv1.weight"], G["conv1.bias"], ws, g.stem_k, g.stem_s, g.stem_p, x3=stem_x3,
                          p16=self.p16 and stem_x3 and 48 < self.h0 <= 60)
        join()
        if saved.get("ps_scope") is not None:
            saved["ps_scope"].release()                    # the saved PS activations / gradients are dead: back to the pool
            saved["blocks"] = []


def param_names(num_blocks: int) -> List[str]:
    """State-dict order of the reference modules (models/PoolResnet.py:70-89)."""
    names = ["conv1.weight", "conv1.bias"]
    for k in range(num_blocks):
        for j in (1, 2):
            names += [f"residual_blocks.{k}.conv{j}.weight", f"residual_blocks.{k}.conv{j}.bias"]
    names += ["out.weight", "out.bias"]
    return names


class ConvStackFn(torch.autograd.Function):
    """Autograd bridge: makes `loss.backward()` work on the model surface (Lightning-style
    callers).  Forward saves activations inside the engine-owned dict; backward returns one
    gradient per parameter."""

    @staticmethod
    def forward(ctx, engine: ConvStack, masks, names, x, *params):
        P = {n: p.detach() for n, p in zip(names, params)}
        y, saved = engine.forward(x.detach(), P, masks, save=True)
        ctx.engine, ctx.saved, ctx.names, ctx.P = engine, saved, names, P
        return y

    @staticmethod
    def backward(ctx, dy):
        G = {n: torch.empty_like(p) for n, p in ctx.P.items()}
        ctx.engine.backward(ctx.saved, dy, ctx.P, G)
        ctx.saved = None
        return (None, None, None, None) + tuple(G[n] for n in ctx.names)
