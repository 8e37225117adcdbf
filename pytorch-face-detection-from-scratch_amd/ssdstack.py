"""SSD conv-stack engine (models/SSD.py:13-255): forward and hand-written backward as a fixed
sequence of HIP kernel launches, assembled from the same bf16x3 primitives as the YOLO stacks.

    stem Conv(3,F,3,s2,p1) -> 9 + 4 SeparableResidualBlocks
        [ (1x1 skip conv iff in != out) ; conv1+lrelu ; conv2+lrelu ; dropout2d(0.25) ; + skip ; (maxpool 2, floor) ]
    after each of the last four blocks: Linear(C,5) on the NHWC map -> rows of the (N,4774,5) output,
    sigmoid on the score, apply_priors.

The 1x1 skip convs and the Linear heads run through the 3x3 kernels with centre-tap weights
(identity activation: slope 1) -- correctness first; they are a few percent of the model's MACs.
A dedicated pointwise kernel is the next step for this row."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import hotpath as hp

F32 = torch.float32
PATCH_SIZES = (60, 30, 15, 7)
HEAD_CP = 16                     # Linear(C,5) padded to 16 output channels (bf16x3 kernels: multiples of 16)


def block_specs(filters: int):
    f = filters
    fe = [(f, 2 * f, True), (2 * f, 2 * f, True)] + [(2 * f, 2 * f, False)] * 6 + [(2 * f, 4 * f, False)]
    specs = [(f"feature_extractor.{k}", i, o, p, -1) for k, (i, o, p) in enumerate(fe)]
    mx = 16 * f
    for i in range(len(PATCH_SIZES)):
        cin = min(4 * f * (2 ** i), mx)
        cout = min(2 * cin, mx)
        specs.append((f"continue_layers.{i}.0", cin, cout, i != 0, i))
    return specs


def param_names(filters: int) -> List[str]:
    """state_dict order of the reference module (stem, all blocks, then the Linear heads)."""
    names = ["input_normalizer.weight", "input_normalizer.bias"]
    for name, ci, co, _, head in block_specs(filters):
        if ci != co:
            names += [name + ".pointwise_conv_skip.weight", name + ".pointwise_conv_skip.bias"]
        names += [name + ".conv1.weight", name + ".conv1.bias", name + ".conv2.weight", name + ".conv2.bias"]
    for head in range(len(PATCH_SIZES)):
        names += [f"extracting_layers.{head}.0.weight", f"extracting_layers.{head}.0.bias"]
    return names


class SSDStack:
    def __init__(self, filters: int, size: int = 480):
        if filters % 16:
            raise ValueError("the SSD engine runs on the bf16x3 kernels: filters must be a multiple of 16")
        if size != 480:
            raise ValueError("patch sizes (60,30,15,7) fix the input at 480x480 (models/SSD.py:108)")
        self.filters, self.size = filters, size
        self.specs = block_specs(filters)
        self.starts = [0]
        for ps in PATCH_SIZES:
            self.starts.append(self.starts[-1] + ps * ps)
        self.P = self.starts[-1]
        self._wpk: Dict[str, torch.Tensor] = {}
        self._key = None
        self._ws: Dict[str, torch.Tensor] = {}
        self.slope = 0.2
        self._pending: list = []                           # same-shape weight gradients awaiting one batched launch
        self.timer = None                                  # convstack.KernelTimer: per-launch HIP events (bench.py's config-4 table)

    def _t(self, kind: str, hk: int, ci: int, co: int):
        if self.timer is None:
            from .convstack import _NOSPAN
            return _NOSPAN
        return self.timer.span(f"{kind}@{hk}x{hk}:{ci}->{co}", 0.0, 0.0)

    # ------------------------------------------------------------------ weights
    def _pack(self, key: str, w3: torch.Tensor):
        co, ci = w3.shape[0], w3.shape[1]
        nf, nb = hp.packed_sizes(co, ci)
        if key + ".f" not in self._wpk:
            self._wpk[key + ".f"] = torch.empty(nf, dtype=F32, device=w3.device)
            self._wpk[key + ".b"] = torch.empty(nb, dtype=F32, device=w3.device)
        hp.pack_conv3x3_weights(w3.contiguous(), self._wpk[key + ".f"], self._wpk[key + ".b"], x3=True)

    def _pack_pointwise(self, key: str, w: torch.Tensor):
        """1x1 skip convs and the Linear heads are dense GEMMs: K-major bf16 hi|lo panels of fdet_pointwise_*_bf16x3."""
        self._wpk[key + ".f"], self._wpk[key + ".b"] = hp.pointwise_pack(w)

    def _ensure_packed(self, P):
        key = tuple((P[k].data_ptr(), P[k]._version) for k in sorted(P) if k.endswith("weight"))
        if key == self._key:
            return
        for name, ci, co, _, head in self.specs:
            if ci != co:
                self._pack_pointwise(name + ".skip", P[name + ".pointwise_conv_skip.weight"])
            self._pack(name + ".conv1", P[name + ".conv1.weight"])
            self._pack(name + ".conv2", P[name + ".conv2.weight"])
            if head >= 0:
                hn = f"extracting_layers.{head}.0"
                self._pack_pointwise(hn, P[hn + ".weight"])
        self._key = key

    def mark_params_dirty(self):
        self._key = None

    def _workspace(self, name, nbytes, dev):
        n = (nbytes + 3) // 4
        t = self._ws.get(name)
        if t is None or t.numel() < n or t.device != dev:
            t = torch.empty(max(n, 4), dtype=F32, device=dev)
            self._ws[name] = t
        return t

    # ------------------------------------------------------------------ forward
    def forward(self, x, P, masks: Optional[Dict[str, torch.Tensor]] = None, save: bool = False):
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, self.size, self.size):
            raise ValueError(f"expected input (N,3,{self.size},{self.size}), got {tuple(x.shape)}")
        x = x.to(F32).contiguous()
        self._ensure_packed(P)
        N, dev, f = x.shape[0], x.device, self.filters
        h0 = self.size // 2
        ws = self._workspace("stem", hp.stem_ws_bytes(N, 3, f, self.size, self.size, 3, 2, 1), dev)
        h = torch.empty(N, f, h0, h0, dtype=F32, device=dev)
        with self._t("stem_fwd", h0, 3, f):
            hp.stem_fwd(x, P["input_normalizer.weight"], P["input_normalizer.bias"], h, ws, 3, 2, 1)
        y = torch.empty(N, self.P, 5, dtype=F32, device=dev)
        saved = {"x": x, "blocks": [], "masks": masks, "heads": {}} if save else None
        for name, ci, co, pool, head in self.specs:
            hk = h.shape[2]
            sc = masks[name] if masks is not None else None
            if ci == co:
                skip = h
            else:
                skip = torch.empty(N, co, hk, hk, dtype=F32, device=dev)
                with self._t("skip1x1_fwd", hk, ci, co):
                    hp.pointwise_fwd(h, self._wpk[name + ".skip.f"], P[name + ".pointwise_conv_skip.bias"], skip)
            a = torch.empty(N, co, hk, hk, dtype=F32, device=dev)
            with self._t("conv3x3_fwd", hk, ci, co):
                hp.conv3x3_fwd(h, self._wpk[name + ".conv1.f"], P[name + ".conv1.bias"], co, y_full=a, slope=self.slope, x3=True)
            ho = hk // 2 if pool else hk
            out = torch.empty(N, co, ho, ho, dtype=F32, device=dev)
            c = torch.empty_like(a) if (pool or save) else None
            if pool:
                with self._t("conv3x3_fwd", hk, co, co):
                    hp.conv3x3_fwd(a, self._wpk[name + ".conv2.f"], P[name + ".conv2.bias"], co, y_full=c, slope=self.slope, x3=True)
                with self._t("tail_fwd", hk, co, co):
                    hp.block_tail_fwd(c, skip, sc, out, 2)
            else:
                with self._t("conv3x3_fwd", hk, co, co):
                    hp.conv3x3_fwd(a, self._wpk[name + ".conv2.f"], P[name + ".conv2.bias"], co, y_full=c, skip=skip,
                                   drop_scale=sc, y_out=out, slope=self.slope, x3=True)
            if save:
                saved["blocks"].append((h, skip, a, c))
            h = out
            if head >= 0:
                hn = f"extracting_layers.{head}.0"
                z = torch.empty(N, 5, ho, ho, dtype=F32, device=dev)       # Linear(C,5) at every position
                with self._t("head", ho, co, 5):
                    hp.pointwise_fwd(h, self._wpk[hn + ".f"], P[hn + ".bias"], z)
                    hp.ssd_head_pack_fwd(z, PATCH_SIZES[head], self.starts[head], y)
                if save:
                    saved["heads"][head] = h
        if save:
            saved["y"] = y
        return y, saved

    # ------------------------------------------------------------------ backward
    def _wgrad(self, xin, dz, dW, db, dev):
        """dW / db (the caller's gradient tensors, written in place) of one 3x3 conv.  Same-shape layers (the six 2F -> 2F
        blocks at 60x60: twelve weight gradients of ~45 us each) are collected and go out in ONE batched launch."""
        N, ci, H, W = xin.shape
        co = dz.shape[1]
        key = (N, ci, co, H, W)
        if self._pending and self._pending[0][0] != key:
            self._flush_wgrads(dev)
        if hp.conv3x3_wgrad_batched_ws_bytes(2, N, ci, co, H, W) > 0:
            self._pending.append((key, xin, dz, dW, db))
            if len(self._pending) == 16:
                self._flush_wgrads(dev)
            return
        x3 = hp.wgrad_x3_supported(N, ci, co, H, W)
        ws = self._workspace("wgrad", hp.conv3x3_wgrad_ws_bytes(N, ci, co, H, W), dev)
        with self._t("conv3x3_wgrad", H, ci, co):
            hp.conv3x3_wgrad(xin, dz, dW, db, ws, x3=x3)

    def _flush_wgrads(self, dev):
        grp, self._pending = self._pending, []
        if not grp:
            return
        N, ci, co, H, W = grp[0][0]
        nb = hp.conv3x3_wgrad_batched_ws_bytes(len(grp), N, ci, co, H, W) if len(grp) > 1 else 0
        if nb == 0:                                        # a single layer, or no batched plan for this many
            ws = self._workspace("wgrad", hp.conv3x3_wgrad_ws_bytes(N, ci, co, H, W), dev)
            for _, xin, dz, dW, db in grp:
                with self._t("conv3x3_wgrad", H, ci, co):
                    hp.conv3x3_wgrad(xin, dz, dW, db, ws, x3=hp.wgrad_x3_supported(N, ci, co, H, W))
            return
        ws = self._workspace("wgrad_batched", nb, dev)
        with self._t(f"conv3x3_wgrad(x{len(grp)})", H, ci, co):
            hp.conv3x3_wgrad_batched([g_[1] for g_ in grp], [g_[2] for g_ in grp], [g_[3] for g_ in grp], [g_[4] for g_ in grp], ws)

    def backward(self, saved, dy, P, G) -> None:
        """dy = d loss / d y (N,4774,5); writes the gradient of every parameter into G[name]."""
        x, masks, y = saved["x"], saved["masks"], saved["y"]
        N, dev = x.shape[0], x.device
        dy = dy.to(F32).contiguous()
        dtrunk = None
        self._pending = []
        for bi in reversed(range(len(self.specs))):
            name, ci, co, pool, head = self.specs[bi]
            hin, skip, a, c = saved["blocks"][bi]
            hk = hin.shape[2]
            ho = hk // 2 if pool else hk
            if head >= 0:                                   # the head reads this block's output
                hn = f"extracting_layers.{head}.0"
                hout = saved["heads"][head]
                dz = torch.empty(N, 5, ho, ho, dtype=F32, device=dev)
                with self._t("head_bwd", ho, co, 5):
                    hp.ssd_head_pack_bwd(dy, y, PATCH_SIZES[head], self.starts[head], dz)
                    hp.pointwise_wgrad(hout, dz, G[hn + ".weight"], G[hn + ".bias"])
                    dout = torch.empty_like(hout)
                    hp.pointwise_dgrad(dz, self._wpk[hn + ".b"], dout, add=dtrunk)
            else:
                dout = dtrunk
            sc = masks[name] if masks is not None else None
            dz2 = torch.empty_like(a)
            with self._t("tail_bwd", hk, co, co):
                if pool:
                    de = torch.empty_like(a)
                    hp.block_tail_bwd(dout, c, skip, sc, dz2, de, 2, self.slope)
                else:
                    de = dout
                    hp.block_tail_bwd(dout, c, None, sc, dz2, None, 1, self.slope)
            self._wgrad(a, dz2, G[name + ".conv2.weight"], G[name + ".conv2.bias"], dev)
            dz1 = torch.empty_like(a)
            with self._t("conv3x3_dgrad", hk, co, co):
                hp.conv3x3_dgrad(dz2, self._wpk[name + ".conv2.b"], co, dz1, act=a, slope=self.slope, x3=True)
            self._wgrad(hin, dz1, G[name + ".conv1.weight"], G[name + ".conv1.bias"], dev)
            if ci == co:
                addt = de
            else:
                with self._t("skip1x1_bwd", hk, ci, co):
                    hp.pointwise_wgrad(hin, de, G[name + ".pointwise_conv_skip.weight"], G[name + ".pointwise_conv_skip.bias"])
                    addt = torch.empty_like(hin)
                    hp.pointwise_dgrad(de, self._wpk[name + ".skip.b"], addt)
            dx = torch.empty_like(hin)
            with self._t("conv3x3_dgrad", hk, co, ci):
                hp.conv3x3_dgrad(dz1, self._wpk[name + ".conv1.b"], ci, dx, add=addt, slope=self.slope, x3=True)
            dtrunk = dx
        self._flush_wgrads(dev)
        ws = self._workspace("stem", hp.stem_ws_bytes(N, 3, self.filters, self.size, self.size, 3, 2, 1), dev)
        with self._t("stem_wgrad", self.size // 2, 3, self.filters):
            hp.stem_wgrad(x, dtrunk, G["input_normalizer.weight"], G["input_normalizer.bias"], ws, 3, 2, 1,
                          x3=hp.stem_k3_wgrad_x3_supported(3, self.filters, self.size, self.size, 3, 2, 1))   # bf16x3 like every other layer of the stack


class SSDStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine: SSDStack, masks, names, x, *params):
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
