"""The remaining optimizers of `base.get_optimizer` (neosr/models/base.py:151-172) on flat HBM arenas:
`Adam` / `NAdam` (torch.optim semantics), `adan` (neosr/optimizers/adan.py), `adamw_sf`
(optimizers/adamw_sf.py) and `adamw_win` (optimizers/adamw_win.py).

One fused elementwise launch per step (`neosr_optim_step`, + `neosr_grad_norm` when the model requests
clipping); the per-step scalar coefficients are computed here in double exactly like the originals, the
model-level `clip_grad_norm_` and the EMA update ride in the same pass (hooks inherited from `AdamW`).
State tensors keep the originals' per-parameter keys as views into flat arenas.
"""

from __future__ import annotations

import ctypes as C
import math

import torch
from torch.optim.optimizer import Optimizer

from neosr_amd import _C
from neosr_amd.hip.nets import arena_layout, flat_grad_of, pack_grads
from neosr_amd.optimizers.adamw import AdamW


class _FlatOptimizer(AdamW):
    """shared plumbing: arenas named `self.ARENAS`, one `neosr_optim_step` per group"""

    ARENAS: tuple[str, ...] = ()
    KIND = 0
    # scalars torch.optim keeps per parameter (`state[p]["step"]`, NAdam's `mu_product`): mirrored as ONE host
    # tensor shared by every state[p] (no device sync) and adopted from a loaded reference state
    PER_PARAM_SCALARS: tuple[str, ...] = ()

    def _init_common(self, params, defaults) -> None:
        Optimizer.__init__(self, params, defaults)
        self._pending_clip, self._grad_scale, self._ema = 0.0, 1.0, None
        self._norm_ws, self.last_grad_norm = None, None
        self._flat: dict[int, dict] = {}

    def _arena_init(self, name: str, p: torch.Tensor, view: torch.Tensor) -> None:
        """state that does not start at zero (e.g. z / x / y = clone of the parameter)"""

    def _ensure_state(self, gi, params, total):
        st = self._flat.get(gi)
        dev = params[0].device
        if st is None or st[self.ARENAS[0]].numel() != total or st[self.ARENAS[0]].device != dev:
            st = {k: torch.zeros(total, device=dev, dtype=torch.float32) for k in self.ARENAS}
            group = self.param_groups[gi]
            shared = {k: torch.zeros((), dtype=torch.float32) for k in self.PER_PARAM_SCALARS}
            for k in self.PER_PARAM_SCALARS:  # loaded torch.optim state: the bias-correction clock continues
                olds = [self.state[p][k] for p in params if k in self.state.get(p, {})]
                if olds and k not in group:  # a file written by torch.optim has it per parameter only
                    group[k] = float(olds[0]) if k != "step" else int(olds[0])
                if k in group:
                    shared[k].fill_(group[k])
            self._shared = getattr(self, "_shared", {})
            self._shared[gi] = shared
            for p, off in zip(params, arena_layout(params)[0]):
                old = self.state.get(p, {})
                new = {k: v for k, v in old.items() if k not in self.ARENAS and k not in shared}
                new.update(shared)
                for k in self.ARENAS:
                    view = st[k][off: off + p.numel()].view(p.shape)
                    if k in old:
                        view.copy_(old[k])
                    else:
                        self._arena_init(k, p, view)
                    new[k] = view
                self.state[p] = new
            self._flat[gi] = st
        return st

    def _coefficients(self, group) -> tuple[list[float], int]:
        raise NotImplementedError

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
                raise _C.NeosrAmdError(f"neosr_amd.{type(self).__name__} needs the parameter group in one flat arena "
                                       "(call flatten_parameters_ before building it)")
            _C.require_device(pflat, "parameter arena")
            gflat = flat_grad_of(params)
            if gflat is None:
                gflat = pack_grads(params)
            total = pflat.numel()
            st = self._ensure_state(gi, params, total)
            coef, flags = self._coefficients(group)
            for k, t in getattr(self, "_shared", {}).get(gi, {}).items():
                t.fill_(group[k])
            if self._norm_ws is None or self._norm_ws.device != pflat.device:
                self._norm_ws = torch.zeros(4200, device=pflat.device, dtype=torch.float32)
            d = _C.OptimDesc(param=pflat.data_ptr(), grad=gflat.data_ptr(), norm_ws=self._norm_ws.data_ptr(), n=total,
                             max_norm=self._pending_clip, ema_decay=0.0, grad_scale=self._grad_scale, kind=self.KIND,
                             flags=flags)
            for i, k in enumerate(self.ARENAS):
                setattr(d, f"s{i}", st[k].data_ptr())
            for i, v in enumerate(coef):
                d.c[i] = v
            if self._ema is not None and len(self.param_groups) == 1:
                ema_arena, decay, first = self._ema
                if ema_arena.numel() != total:
                    raise _C.NeosrAmdError("EMA arena size does not match the parameter arena")
                d.ema = ema_arena.data_ptr()
                d.ema_decay = -1.0 if first else decay
            _C.check(lib.neosr_optim_step(C.byref(d), _C.stream_ptr()), "neosr_optim_step")
            _C.params_changed()
            if self._pending_clip > 0:
                self.last_grad_norm = self._norm_ws[0]
        self._pending_clip = 0.0
        self._ema = None
        return loss


def _check_basic(lr, eps, betas) -> None:
    if not lr >= 0.0:
        raise ValueError(f"Invalid learning rate: {lr}")
    if not eps >= 0.0:
        raise ValueError(f"Invalid epsilon value: {eps}")
    for i, b in enumerate(betas):
        if not 0.0 <= b < 1.0:
            raise ValueError(f"Invalid beta parameter at index {i}: {b}")


class Adam(_FlatOptimizer):
    """torch.optim.Adam (L2 weight decay added to the gradient)."""

    ARENAS, KIND = ("exp_avg", "exp_avg_sq"), _C.OPT_ADAM
    PER_PARAM_SCALARS = ("step",)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **kwargs):  # noqa: ARG002
        _check_basic(lr, eps, betas)
        if amsgrad:
            raise NotImplementedError("Adam: amsgrad has no HIP path")
        self._init_common(params, {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay})

    def _coefficients(self, group):
        group["step"] = group.get("step", 0) + 1
        t, (b1, b2) = group["step"], group["betas"]
        return [b1, b2, group["eps"], group["weight_decay"], group["lr"] / (1 - b1**t), math.sqrt(1 - b2**t)], 0


class NAdam(_FlatOptimizer):
    """torch.optim.NAdam (momentum_decay schedule, L2 weight decay)."""

    ARENAS, KIND = ("exp_avg", "exp_avg_sq"), _C.OPT_NADAM
    PER_PARAM_SCALARS = ("step", "mu_product")

    def __init__(self, params, lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, momentum_decay=4e-3,
                 decoupled_weight_decay=False, **kwargs):  # noqa: ARG002
        _check_basic(lr, eps, betas)
        if decoupled_weight_decay:
            raise NotImplementedError("NAdam: decoupled_weight_decay has no HIP path")
        self._init_common(params, {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay,
                                   "momentum_decay": momentum_decay})

    def _coefficients(self, group):
        group["step"] = group.get("step", 0) + 1
        t, (b1, b2), md, lr = group["step"], group["betas"], group["momentum_decay"], group["lr"]
        mu = b1 * (1.0 - 0.5 * (0.96 ** (t * md)))
        mu_next = b1 * (1.0 - 0.5 * (0.96 ** ((t + 1) * md)))
        group["mu_product"] = group.get("mu_product", 1.0) * mu
        mp = group["mu_product"]
        return [b1, b2, group["eps"], group["weight_decay"], lr * (1.0 - mu) / (1.0 - mp), math.sqrt(1 - b2**t),
                lr * mu_next / (1.0 - mp * mu_next)], 0


class adan(_FlatOptimizer):
    """neosr/optimizers/adan.py (Adan with optional proximal weight decay)."""

    ARENAS, KIND = ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad"), _C.OPT_ADAN

    def __init__(self, params, lr=1e-3, betas=(0.98, 0.92, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0,
                 no_prox=False, foreach=True, **kwargs):  # noqa: ARG002
        _check_basic(lr, eps, betas)
        if not max_grad_norm >= 0.0:
            raise ValueError(f"Invalid Max grad norm: {max_grad_norm}")
        if max_grad_norm > 0.0:
            raise NotImplementedError("adan: max_grad_norm > 0 has no HIP path; the model-level grad_clip is fused")
        self._init_common(params, {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay,
                                   "max_grad_norm": max_grad_norm, "no_prox": no_prox, "foreach": foreach})

    def _coefficients(self, group):
        group["step"] = group.get("step", 0) + 1
        t, (b1, b2, b3), lr = group["step"], group["betas"], group["lr"]
        bc1, bc2, bc3 = 1.0 - b1**t, 1.0 - b2**t, 1.0 - b3**t
        return [b1, b2, b3, group["eps"], lr * group["weight_decay"], lr / bc1, lr * b2 / bc2, math.sqrt(bc3),
                1.0 if group["no_prox"] else 0.0], int(t == 1)


class adamw_sf(_FlatOptimizer):
    """neosr/optimizers/adamw_sf.py (Schedule-Free AdamW): needs `.train()` / `.eval()` like the original."""

    ARENAS, KIND = ("exp_avg_sq", "z"), _C.OPT_ADAMW_SF

    def __init__(self, params, lr=0.0025, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, warmup_steps=0, r=0.0,
                 weight_lr_power=2.0, foreach=True, schedule_free=True, **kwargs):  # noqa: ARG002
        self._init_common(params, {"lr": lr, "betas": betas, "eps": eps, "r": r, "k": 0,
                                   "warmup_steps": warmup_steps, "train_mode": True, "weight_sum": 0.0, "lr_max": -1.0,
                                   "weight_lr_power": weight_lr_power, "weight_decay": weight_decay, "foreach": foreach})

    def _arena_init(self, name, p, view):
        if name == "z":
            view.copy_(p.detach())

    def _lerp_params(self, weight_of_beta1) -> None:
        lib = _C.load()
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            if st is None:
                continue
            pflat, _ = self._group_arena(group)
            _C.check(lib.neosr_lerp(pflat.data_ptr(), st["z"].data_ptr(), pflat.numel(), weight_of_beta1(group["betas"][0]),
                                    _C.stream_ptr()), "neosr_lerp")
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

    def _coefficients(self, group):
        k, (b1, b2) = group["k"], group["betas"]
        ws_ = group["warmup_steps"]
        sched = (k + 1) / ws_ if k < ws_ else 1.0
        lr = group["lr"] * sched * math.sqrt(1 - b2 ** (k + 1))
        lr_max = group["lr_max"] = max(lr, group["lr_max"])
        weight = ((k + 1) ** group["r"]) * (lr_max ** group["weight_lr_power"])
        weight_sum = group["weight_sum"] = group["weight_sum"] + weight
        try:
            ckp1 = weight / weight_sum
        except ZeroDivisionError:
            ckp1 = 0
        if not group["train_mode"]:
            raise ValueError("Not in train mode!")
        group["k"] = k + 1
        return [b2, group["eps"], group["weight_decay"], ckp1, lr, lr * (b1 * (1 - ckp1) - 1)], 0


class adamw_win(_FlatOptimizer):
    """neosr/optimizers/adamw_win.py (AdamW with Win / Win2 acceleration)."""

    ARENAS, KIND = ("exp_avg", "exp_avg_sq", "x", "y"), _C.OPT_ADAMW_WIN

    def __init__(self, params, lr=5e-4, betas=(0.98, 0.999), reckless_steps=(2.0, 8.0), eps=1e-8, weight_decay=0.02,
                 amsgrad=False, max_grad_norm=0.0, acceleration_mode="win2", **kwargs):  # noqa: ARG002
        _check_basic(lr, eps, betas)
        if reckless_steps[0] < 0.0 or reckless_steps[1] < 0.0:
            raise ValueError(f"Invalid reckless_steps parameter: {reckless_steps}")
        if amsgrad or max_grad_norm > 1e-8:
            raise NotImplementedError("adamw_win: amsgrad / max_grad_norm have no HIP path (model-level grad_clip is fused)")
        self._init_common(params, {"lr": lr, "betas": betas, "reckless_steps": tuple(reckless_steps), "eps": eps,
                                   "weight_decay": weight_decay, "amsgrad": amsgrad, "max_grad_norm": max_grad_norm,
                                   "acceleration_mode": acceleration_mode})

    def _arena_init(self, name, p, view):
        if name in ("x", "y"):
            view.copy_(p.detach())

    def _coefficients(self, group):
        group["step"] = group.get("step", 0) + 1
        t, (b1, b2) = group["step"], group["betas"]
        mode = group["acceleration_mode"]
        flags = 2 if mode == "win2" else (1 if "win" in mode else 0)
        b3, b4 = group["reckless_steps"]
        return [b1, b2, group["eps"], group["weight_decay"], group["lr"], 1 - b1**t, math.sqrt(1 - b2**t), b3, b4], flags
