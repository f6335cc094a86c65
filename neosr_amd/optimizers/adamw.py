"""AdamW on a flat HBM arena: global-norm clip + decoupled-decay Adam + EMA in two kernel launches.

Replaces, for `optim_g.type = "adamw"` (neosr/models/base.py:151-172), the sequence
`clip_grad_norm_` (image.py:533-544) -> `torch.optim.AdamW.step` -> `AveragedModel.update_parameters`
(image.py:661-662) — ~20 full sweeps over the parameters in the reference — by
`neosr_grad_norm` + `neosr_adamw_step`.  Keeps torch.optim.Optimizer's interface and state-dict
layout (`state[p] = {step, exp_avg, exp_avg_sq}`, the moments being views into flat arenas) so
`.state` checkpoints round-trip.
"""

from __future__ import annotations

import ctypes as C

import torch
from torch.optim.optimizer import Optimizer

from neosr_amd import _C
from neosr_amd.hip.nets import arena_layout, flat_grad_of, flat_view_of, group_arena, pack_grads


class AdamW(Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, **kwargs) -> None:  # noqa: ARG002
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        defaults = {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}
        super().__init__(params, defaults)
        self._pending_clip: float = 0.0
        self._grad_scale: float = 1.0
        self._ema: tuple[torch.Tensor, float, bool] | None = None
        self._norm_ws: torch.Tensor | None = None
        self.last_grad_norm: torch.Tensor | None = None
        self._flat: dict[int, dict] = {}  # per-group flat moment arenas (not part of state_dict)

    # -- hooks used by the model to fuse work into the step ---------------------------------
    def set_clip(self, max_norm: float) -> None:
        """Request `clip_grad_norm_(params, max_norm)` semantics inside the next `step()`."""
        self._pending_clip = float(max_norm)

    def set_grad_scale(self, scale: float) -> None:
        """Scale grads on the fly (1/world_size after a SUM all-reduce)."""
        self._grad_scale = float(scale)

    def set_ema(self, ema_arena: torch.Tensor, decay: float, first: bool) -> None:
        """Fuse the EMA update of a flat shadow arena into the next `step()`."""
        self._ema = (ema_arena, float(decay), bool(first))

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self._flat = {}  # moments are re-adopted from self.state into fresh arenas on next step

    # -- state ---------------------------------------------------------------------------
    def _group_arena(self, group):
        """(param_flat, params) if the group's params sit in one buffer at the arena layout."""
        return group_arena(self.__dict__.setdefault("_neosr_group_cache", {}), group["params"])

    def _ensure_state(self, gi, params, total):
        st = self._flat.get(gi)
        dev = params[0].device
        if st is None or st["exp_avg"].numel() != total or st["exp_avg"].device != dev:
            m = torch.zeros(total, device=dev, dtype=torch.float32)
            v = torch.zeros(total, device=dev, dtype=torch.float32)
            step0 = 0
            shared_step = torch.zeros((), device="cpu", dtype=torch.float32)  # one host scalar
            for p, off in zip(params, arena_layout(params)[0]):
                n = p.numel()
                old = self.state.get(p, {})
                if "exp_avg" in old:  # resumed from a checkpoint: adopt its moments
                    m[off : off + n].copy_(old["exp_avg"].reshape(-1))
                    v[off : off + n].copy_(old["exp_avg_sq"].reshape(-1))
                    step0 = int(old["step"])
                self.state[p] = {
                    "step": shared_step,
                    "exp_avg": m[off : off + n].view(p.shape),
                    "exp_avg_sq": v[off : off + n].view(p.shape),
                }
            shared_step.fill_(step0)
            st = {"exp_avg": m, "exp_avg_sq": v, "step": shared_step, "step_int": step0}
            self._flat[gi] = st
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _C.load()
        for gi, group in enumerate(self.param_groups):
            pflat, params = self._group_arena(group)
            if pflat is None:
                raise _C.NeosrAmdError(
                    "neosr_amd.AdamW needs the parameter group in one flat arena "
                    "(call module.flat_parameters() / flatten_parameters_ before building it)")
            _C.require_device(pflat, "parameter arena")
            gflat = flat_grad_of(params)
            if gflat is None:  # grads produced elsewhere (not by our plans): pack them once
                gflat = pack_grads(params)
            total = pflat.numel()
            st = self._ensure_state(gi, params, total)
            st["step_int"] += 1
            step = st["step_int"]
            st["step"].fill_(step)  # host tensor shared by every state[p]["step"]: no device sync
            if self._norm_ws is None or self._norm_ws.device != pflat.device:
                self._norm_ws = torch.zeros(4200, device=pflat.device, dtype=torch.float32)
            d = _C.AdamWDesc()
            d.param, d.grad = pflat.data_ptr(), gflat.data_ptr()
            d.exp_avg, d.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            d.norm_ws = self._norm_ws.data_ptr()
            d.n = total
            d.lr, (d.beta1, d.beta2) = group["lr"], group["betas"]
            d.eps, d.weight_decay = group["eps"], group["weight_decay"]
            d.max_norm = self._pending_clip
            d.grad_scale = self._grad_scale
            d.step = step
            d.ema = None
            d.ema_decay = 0.0
            if self._ema is not None and len(self.param_groups) == 1:
                ema_arena, decay, first = self._ema
                if ema_arena.numel() != total:
                    raise _C.NeosrAmdError("EMA arena size does not match the parameter arena")
                d.ema = ema_arena.data_ptr()
                d.ema_decay = -1.0 if first else decay
            _C.check(lib.neosr_adamw_step(C.byref(d), _C.stream_ptr()), "neosr_adamw_step")
            _C.params_changed()
            if self._pending_clip > 0:
                self.last_grad_norm = self._norm_ws[0]
        self._pending_clip = 0.0
        self._ema = None
        return loss
