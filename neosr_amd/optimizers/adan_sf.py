"""Schedule-Free Adan on flat HBM arenas (drop-in for neosr/optimizers/adan_sf.py:10-330).

Same constructor, same `train()` / `eval()` contract, same per-parameter state keys (`exp_avg`,
`exp_avg_sq`, `exp_avg_diff`, `z`, `neg_pre_grad`: views into flat arenas) and the same group
bookkeeping (`step`, `weight_sum`, `lr_max`, `train_mode`).  `step()` is `neosr_grad_norm` +
`neosr_adan_sf_step`: the model-level `clip_grad_norm_`, the 17 `_foreach_*` sweeps of
`_multi_tensor_adan` and the EMA update in one pass over 7 arenas.
"""

from __future__ import annotations

import ctypes as C
import math

import torch
from torch.optim.optimizer import Optimizer

from neosr_amd import _C
from neosr_amd.hip.nets import arena_layout, flat_grad_of, flat_view_of, pack_grads
from neosr_amd.optimizers.adamw import AdamW

_ARENAS = ("exp_avg", "exp_avg_sq", "exp_avg_diff", "z", "neg_pre_grad")


class adan_sf(AdamW):
    def __init__(self, params, lr: float = 1.6e-3, betas=(0.98, 0.92, 0.99), eps: float = 1e-8,
                 weight_decay: float = 0.02, max_grad_norm: float = 0.0, warmup_steps: int = 0, r: float = 0.0,
                 weight_lr_power: float = 2.0, schedule_free: bool = True, **kwargs) -> None:  # noqa: ARG002
        if not max_grad_norm >= 0.0:
            raise ValueError(f"Invalid Max grad norm: {max_grad_norm}")
        if max_grad_norm > 0.0:
            raise NotImplementedError("adan_sf: max_grad_norm > 0 (the optimizer's own clip) has no HIP path; "
                                      "the model-level grad_clip is fused instead")
        if not lr >= 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not eps >= 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        for i in range(3):
            if not 0.0 <= betas[i] < 1.0:
                raise ValueError(f"Invalid beta parameter at index {i}: {betas[i]}")
        defaults = {"lr": lr, "betas": betas, "eps": eps, "r": r, "weight_decay": weight_decay,
                    "max_grad_norm": max_grad_norm, "warmup_steps": warmup_steps, "train_mode": True,
                    "weight_sum": 0.0, "lr_max": -1.0, "weight_lr_power": weight_lr_power,
                    "schedule_free": schedule_free}
        Optimizer.__init__(self, params, defaults)
        self._pending_clip = 0.0
        self._grad_scale = 1.0
        self._ema = None
        self._norm_ws = None
        self.last_grad_norm = None
        self._flat: dict[int, dict] = {}

    def __setstate__(self, state) -> None:
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("schedule_free", True)

    # -- y <-> x switch of Schedule-Free (adan_sf.py:112-136) ------------------------------------
    def _lerp_params(self, weight_of_beta1) -> None:
        lib = _C.load()
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            if st is None:
                continue  # no step taken yet: no z
            pflat, _ = self._group_arena(group)
            _C.check(lib.neosr_lerp(pflat.data_ptr(), st["z"].data_ptr(), pflat.numel(),
                                    weight_of_beta1(group["betas"][0]), _C.stream_ptr()), "neosr_lerp")
            _C.params_changed()

    @torch.no_grad()
    def eval(self) -> None:
        todo = [g for g in self.param_groups if g["train_mode"]]
        if todo:
            self._lerp_params(lambda b1: 1 - 1 / b1)
        for g in todo:
            g["train_mode"] = False

    @torch.no_grad()
    def train(self) -> None:
        todo = [g for g in self.param_groups if not g["train_mode"]]
        if todo:
            self._lerp_params(lambda b1: 1 - b1)
        for g in todo:
            g["train_mode"] = True

    # -- state -------------------------------------------------------------------------------
    def _ensure_state(self, gi, params, total):
        st = self._flat.get(gi)
        dev = params[0].device
        if st is None or st["exp_avg"].numel() != total or st["exp_avg"].device != dev:
            st = {k: torch.zeros(total, device=dev, dtype=torch.float32) for k in _ARENAS}
            for p, off in zip(params, arena_layout(params)[0]):
                n = p.numel()
                old = self.state.get(p, {})
                new = {}
                for k in _ARENAS:
                    view = st[k][off: off + n].view(p.shape)
                    if k in old:  # resumed from a checkpoint
                        view.copy_(old[k])
                    elif k == "z":
                        view.copy_(p.detach())  # state["z"] = torch.clone(p)
                    new[k] = view
                self.state[p] = new
            self._flat[gi] = st
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = 0.0
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _C.load()
        for gi, group in enumerate(self.param_groups):
            pflat, params = self._group_arena(group)
            if pflat is None:
                raise _C.NeosrAmdError("neosr_amd.adan_sf needs the parameter group in one flat arena "
                                       "(call flatten_parameters_ before building it)")
            _C.require_device(pflat, "parameter arena")
            gflat = flat_grad_of(params)
            if gflat is None:
                gflat = pack_grads(params)
            total = pflat.numel()
            st = self._ensure_state(gi, params, total)
            group["step"] = group.get("step", 0) + 1
            step = group["step"]
            beta1, beta2, beta3 = group["betas"]
            ckp1 = 0.0
            if group["schedule_free"]:
                # adan_sf.py:187-210, host-side scalar state
                ws_ = group["warmup_steps"]
                sched = step / ws_ if step < ws_ else 1.0
                lr_eff = group["lr"] * sched * math.sqrt(1.0 - beta3 ** step)
                lr_max = group["lr_max"] = max(lr_eff, group["lr_max"])
                weight = (step ** group["r"]) * (lr_max ** group["weight_lr_power"])
                weight_sum = group["weight_sum"] = group["weight_sum"] + weight
                try:
                    ckp1 = weight / weight_sum
                except ZeroDivisionError:
                    ckp1 = 0
                if not group["train_mode"]:
                    raise ValueError("Not in train mode!")
            if self._norm_ws is None or self._norm_ws.device != pflat.device:
                self._norm_ws = torch.zeros(4200, device=pflat.device, dtype=torch.float32)
            d = _C.AdanDesc(param=pflat.data_ptr(), grad=gflat.data_ptr(), exp_avg=st["exp_avg"].data_ptr(),
                            exp_avg_sq=st["exp_avg_sq"].data_ptr(), exp_avg_diff=st["exp_avg_diff"].data_ptr(),
                            z=st["z"].data_ptr(), neg_pre_grad=st["neg_pre_grad"].data_ptr(), ema=None,
                            norm_ws=self._norm_ws.data_ptr(), n=total, lr=group["lr"], beta1=beta1, beta2=beta2,
                            beta3=beta3, eps=group["eps"], weight_decay=group["weight_decay"], ckp1=ckp1,
                            max_norm=self._pending_clip, ema_decay=0.0, grad_scale=self._grad_scale, step=step,
                            first_step=0, schedule_free=int(bool(group["schedule_free"])))
            if self._ema is not None and len(self.param_groups) == 1:
                ema_arena, decay, first = self._ema
                if ema_arena.numel() != total:
                    raise _C.NeosrAmdError("EMA arena size does not match the parameter arena")
                d.ema = ema_arena.data_ptr()
                d.ema_decay = -1.0 if first else decay
            _C.check(lib.neosr_adan_sf_step(C.byref(d), _C.stream_ptr()), "neosr_adan_sf_step")
            _C.params_changed()
            if self._pending_clip > 0:
                self.last_grad_norm = self._norm_ws[0]
        self._pending_clip = 0.0
        self._ema = None
        return loss
