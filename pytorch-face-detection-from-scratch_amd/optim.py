"""Fused Adam on one flat parameter buffer (single HIP launch per step).

Replaces `SAMSGD(Adam)` + `torch.optim._multi_tensor.Adam.step`, models/ModelMeta.py:12-82.
The reference's SAM wrapper perturbs the weights and removes the perturbation again without a
second backward (its closure never calls .backward(), :121-131), so the update it performs IS
plain Adam on the batch-sum gradient; the two stale forward passes and the w+e-e rounding are
deliberately not reproduced.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import hotpath as hp


class FlatSpace:
    """All parameters (and their gradients / Adam moments) as views into flat fp32 buffers.
    Every view starts on a 16-byte boundary so the elementwise kernels can use 16 B lanes."""

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = list(params)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        self.flat = self.grad = self.exp_avg = self.exp_avg_sq = None
        self._build()

    def _build(self):
        dev = self.params[0].device
        flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).float())
        keep_state = self.exp_avg is not None and self.exp_avg.numel() == self.total
        self.flat = flat
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.exp_avg = self.exp_avg.to(dev) if keep_state else torch.zeros_like(flat)
        self.exp_avg_sq = self.exp_avg_sq.to(dev) if keep_state else torch.zeros_like(flat)
        for p, o in zip(self.params, self.offsets):
            p.data = flat[o:o + p.numel()].view(p.shape)

    def ensure(self) -> bool:
        """Re-flatten if something (module.to(), load_state_dict with assign) re-homed a
        parameter.  Returns True when a rebuild happened."""
        base = self.flat.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != base + 4 * o or p.device != self.flat.device:
                self._build()
                return True
        return False

    def view(self, buf: torch.Tensor, i: int) -> torch.Tensor:
        p, o = self.params[i], self.offsets[i]
        return buf[o:o + p.numel()].view(p.shape)

    def gather_autograd_grads(self) -> None:
        """Copy p.grad (autograd path) into the flat gradient buffer."""
        for i, p in enumerate(self.params):
            gv = self.view(self.grad, i)
            if p.grad is None:
                gv.zero_()
            elif p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)


class SAMSGD(torch.optim.Optimizer):
    """Name and constructor of the reference's optimiser (models/ModelMeta.py:12-41); the
    update is Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) in one fused launch."""

    def __init__(self, params: Iterable, lr: float, rho: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8):
        if rho <= 0:
            raise ValueError(f"Invalid neighborhood size: {rho}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) > 1:
            raise ValueError("Not supported")
        self.param_groups[0]["rho"] = rho
        self.closure = None
        self.space: Optional[FlatSpace] = None
        self.step_count = 0
        self.on_params_updated = None        # e.g. ConvStack.mark_params_dirty

    def set_closure_fn(self, closure):
        self.closure = closure               # the reference's SAM closure (two stale forwards, Q18): kept, never called

    def _space(self) -> FlatSpace:
        if self.space is None:
            self.space = FlatSpace(self.param_groups[0]["params"])
        else:
            self.space.ensure()
        return self.space

    # ------------------------------------------------------------------ checkpoint round trip
    def state_dict(self):
        """torch.optim.Adam layout: state[i] = {step, exp_avg, exp_avg_sq} per parameter (copies of the views
        into the flat moment buffers), so checkpoints interchange with the reference's `optimizer_states`
        (its SAMSGD subclasses Adam, models/ModelMeta.py:12) and a resumed run continues with its moments
        and bias correction."""
        sd = super().state_dict()
        if self.space is not None and self.step_count > 0:
            sp = self.space
            sd["state"] = {i: {"step": torch.tensor(float(self.step_count)),
                               "exp_avg": sp.view(sp.exp_avg, i).clone(),
                               "exp_avg_sq": sp.view(sp.exp_avg_sq, i).clone()}
                           for i in range(len(sp.params))}
        return sd

    def load_state_dict(self, state_dict):
        state = state_dict.get("state", {}) or {}
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        g = self.param_groups[0]
        g.setdefault("rho", 0.05)
        sp = self._space()
        sp.exp_avg.zero_()
        sp.exp_avg_sq.zero_()
        self.step_count = 0
        if not state:
            return
        by_idx = {int(k): v for k, v in state.items()}
        if sorted(by_idx) != list(range(len(sp.params))):
            raise ValueError(f"optimizer state holds {len(by_idx)} entries for {len(sp.params)} parameters")
        steps = set()
        with torch.no_grad():
            for i, p in enumerate(sp.params):
                st = by_idx[i]
                for key, buf in (("exp_avg", sp.exp_avg), ("exp_avg_sq", sp.exp_avg_sq)):
                    t = st[key]
                    if tuple(t.shape) != tuple(p.shape):
                        raise ValueError(f"optimizer state {key}[{i}] has shape {tuple(t.shape)}, parameter {tuple(p.shape)}")
                    sp.view(buf, i).copy_(t.to(device=buf.device, dtype=torch.float32))
                steps.add(int(float(st["step"])))            # int in torch 1.10, float tensor later
        if len(steps) != 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): one flat Adam step cannot resume that")
        self.step_count = steps.pop()

    def step(self, closure=None, grads_in_flat: bool = False, grad_scale: float = 1.0):
        """`closure` (torch.optim contract / Lightning's automatic optimisation): re-evaluates the model and
        runs backward; it is called with grad enabled BEFORE the gradients are gathered, and its loss returned."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._step(grads_in_flat, grad_scale)
        return loss

    @torch.no_grad()
    def _step(self, grads_in_flat: bool, grad_scale: float):
        sp = self._space()
        if not grads_in_flat:
            sp.gather_autograd_grads()
        g = self.param_groups[0]
        self.step_count += 1
        hp.adam_step(sp.flat, sp.grad, sp.exp_avg, sp.exp_avg_sq, self.step_count, lr=g["lr"], beta1=g["betas"][0],
                     beta2=g["betas"][1], eps=g["eps"], grad_scale=grad_scale)
        if self.on_params_updated is not None:
            self.on_params_updated()

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
