"""Friendly Sharpness-Aware Minimization on flat HBM arenas (neosr/optimizers/fsam.py).

`step(closure, current_iter)` = `first_step` (momentum-corrected gradient, adaptive norm, climb to
w + e(w): `neosr_fsam_first_step`, two launches over the arenas instead of ~12 sweeps per tensor), the
closure again at the perturbed weights, then `second_step` (restore w, the base optimizer's fused step).
Like the reference, the base optimizer is a SECOND instance with its own state (image.py:322-347): the
model's `optimizer_g` is not stepped while SAM is active.  `state[p] = {momentum, old_p}` are views into
flat arenas.
"""

from __future__ import annotations

import ctypes as C

import torch
from torch.optim.optimizer import Optimizer

from neosr_amd import _C
from neosr_amd.hip.nets import arena_layout, flat_grad_of, flat_view_of, pack_grads


class fsam(Optimizer):
    def __init__(self, params, base_optimizer, rho: float = 0.5, sigma: float = 1.0, lmbda: float = 0.9,
                 adaptive: bool = True, **kwargs) -> None:
        assert rho >= 0.0, f"Invalid rho, should be non-negative: {rho}"
        defaults = dict(rho=rho, adaptive=adaptive, **kwargs)
        super().__init__(params, defaults)
        self.base_optimizer = base_optimizer(self.param_groups, **kwargs)
        self.param_groups = self.base_optimizer.param_groups
        self.defaults.update(self.base_optimizer.defaults)
        self.sigma = sigma
        self.lmbda = lmbda
        self._grad_scale = 1.0
        self._flat: dict[int, dict] = {}
        self._norm_ws: torch.Tensor | None = None
        self.last_grad_norm: torch.Tensor | None = None

    # hooks the model uses for the fused work of the base step (see AdamW)
    def set_grad_scale(self, scale: float) -> None:
        self._grad_scale = float(scale)
        self.base_optimizer.set_grad_scale(scale)

    def set_clip(self, max_norm: float) -> None:  # noqa: ARG002 - no clipping while SAM is active (image.py:533-544)
        return

    def set_ema(self, ema_arena: torch.Tensor, decay: float, first: bool) -> None:
        self.base_optimizer.set_ema(ema_arena, decay, first)

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)
        self.base_optimizer.param_groups = self.param_groups
        self._flat = {}

    def _arenas(self, gi, params, total):
        st = self._flat.get(gi)
        dev = params[0].device
        if st is None or st["momentum"].numel() != total or st["momentum"].device != dev:
            st = {k: torch.zeros(total, device=dev, dtype=torch.float32) for k in ("momentum", "old_p")}
            st["first"] = True
            for p, off in zip(params, arena_layout(params)[0]):
                old = self.state.get(p, {})
                new = {}
                for k in ("momentum", "old_p"):
                    view = st[k][off: off + p.numel()].view(p.shape)
                    if k in old:
                        view.copy_(old[k])
                        if k == "momentum":
                            st["first"] = False
                    new[k] = view
                self.state[p] = new
            self._flat[gi] = st
        return st

    @torch.no_grad()
    def first_step(self, zero_grad: bool = False) -> None:
        lib = _C.load()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.requires_grad and p.grad is not None]
            if not params:
                continue
            pflat = flat_view_of([p.data for p in params])
            if pflat is None:
                raise _C.NeosrAmdError("neosr_amd.fsam needs the parameter group in one flat arena "
                                       "(call flatten_parameters_ before building it)")
            _C.require_device(pflat, "parameter arena")
            gflat = flat_grad_of(params)
            if gflat is None:
                gflat = pack_grads(params)
            total = pflat.numel()
            st = self._arenas(gi, params, total)
            if self._norm_ws is None or self._norm_ws.device != pflat.device:
                self._norm_ws = torch.zeros(4 + 1024, device=pflat.device, dtype=torch.float32)
            d = _C.FsamDesc(param=pflat.data_ptr(), grad=gflat.data_ptr(), momentum=st["momentum"].data_ptr(),
                            old_p=st["old_p"].data_ptr(), norm_ws=self._norm_ws.data_ptr(), n=total,
                            rho=group["rho"], sigma=self.sigma, lmbda=self.lmbda, grad_scale=self._grad_scale,
                            first=int(st["first"]), adaptive=int(bool(group["adaptive"])))
            _C.check(lib.neosr_fsam_first_step(C.byref(d), _C.stream_ptr()), "neosr_fsam_first_step")
            _C.params_changed()
            st["first"] = False
            self.last_grad_norm = self._norm_ws[0]
        if zero_grad:
            self.zero_grad(set_to_none=True)

    @torch.no_grad()
    def second_step(self, zero_grad: bool = False) -> None:
        for gi, group in enumerate(self.param_groups):
            st = self._flat.get(gi)
            if st is None:
                continue
            params = [p for p in group["params"] if p.requires_grad]
            pflat = flat_view_of([p.data for p in params])
            pflat.copy_(st["old_p"])  # back to "w" from "w + e(w)"
            _C.params_changed()
        self.base_optimizer.step()  # the actual sharpness-aware update
        if zero_grad:
            self.zero_grad(set_to_none=True)

    @torch.no_grad()
    def step(self, closure=None, current_iter: int | None = None):
        assert closure is not None, "Sharpness Aware Minimization requires closure, but it was not provided"
        closure = torch.enable_grad()(closure)  # the closure does a full forward-backward pass
        self.first_step(zero_grad=True)
        closure(current_iter)
        self.second_step()
