"""Pixel losses (drop-in for neosr/losses/basic_loss.py).

``L1Loss`` / ``MSELoss`` / ``HuberLoss`` with reduction "mean" or "sum" and ``chc_loss`` (with or without its
cosine-similarity term) run on the HIP reduction kernels (fixed-order two-stage sums, run-to-run deterministic).
reduction="none" returns a per-element map that `image.closure` could not back-propagate in the reference either
(`l_g_total.backward()` needs a scalar); it is rejected loudly rather than computed elsewhere.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from neosr_amd import _C

from neosr_amd.hip.layers import ChcCosLoss, ChcLoss
from neosr_amd.hip.nets import L1LossFunction
from neosr_amd.utils.registry import LOSS_REGISTRY

_reduction_modes = ["none", "mean", "sum"]


@LOSS_REGISTRY.register()
class L1Loss(nn.Module):
    """L1 (MAE) loss, `loss_weight * mean|pred - target|` (basic_loss.py:24-53)."""

    def __init__(self, loss_weight: float = 1.0, reduction: str = "mean") -> None:
        super().__init__()
        if reduction not in _reduction_modes:
            msg = f"Unsupported reduction mode: {reduction}. Supported ones are: {_reduction_modes}"
            raise ValueError(msg)
        if reduction == "none":
            msg = "neosr_amd L1Loss: reduction='none' has no HIP kernel (the training loop needs a scalar loss)"
            raise NotImplementedError(msg)
        self.loss_weight = loss_weight
        self.reduction = reduction

    def forward(self, pred: Tensor, target: Tensor, **kwargs) -> Tensor:  # noqa: ARG002
        # "sum" = numel * mean: the same kernels with the weight scaled
        w = self.loss_weight * (pred.numel() if self.reduction == "sum" else 1)
        return L1LossFunction.apply(pred, target, w)


@LOSS_REGISTRY.register()
class chc_loss(nn.Module):
    """Clipped pseudo-Huber (+ cosine term) loss (basic_loss.py:132-219).

    `loss_weight * mean(clamp(t + loss_lambda * (1 - cos_sim).mean(), clip_min, clip_max))` with
    `t = |d|` ("l1") or `sqrt(d^2 + 1e-12)` ("huber"), cos_sim over dim 1 of the (N, C, H, W) tensors.
    loss_lambda = 0 (the class default and vgg_perceptual_loss.py:144) skips the cosine pass: the term then
    shifts nothing and receives no gradient.  `reduction` is accepted and, as in the reference's forward
    (basic_loss.py:192-219 always takes `torch.mean`), has no effect."""

    def __init__(self, loss_weight: float = 1.0, reduction: str = "mean", criterion: str = "huber",
                 loss_lambda: float = 0, clip_min: float = 0.003921, clip_max: float = 0.996078) -> None:
        super().__init__()
        if reduction not in {"none", "mean", "sum"}:
            msg = f"Unsupported reduction mode: {reduction}. Supported ones are: {_reduction_modes}"
            raise ValueError(msg)
        if criterion not in {"l1", "huber"}:
            raise NotImplementedError(f"{criterion} not implemented.")
        self.loss_weight, self.criterion = loss_weight, criterion
        self.loss_lambda, self.clip_min, self.clip_max = loss_lambda, clip_min, clip_max

    def forward(self, pred: Tensor, target: Tensor, **kwargs) -> Tensor:  # noqa: ARG002
        if self.loss_lambda != 0:
            return ChcCosLoss.apply(pred, target, self.criterion == "huber", float(self.clip_min),
                                    float(self.clip_max), float(self.loss_lambda), float(self.loss_weight))
        return ChcLoss.apply(pred, target, 1.0, self.criterion == "huber", float(self.clip_min),
                             float(self.clip_max), float(self.loss_weight))


class _PointwiseLoss(torch.autograd.Function):
    """loss_weight * mean(term(pred - target)), term = d^2 (MSE) or Huber(delta); `neosr_pointwise_loss_*`."""

    @staticmethod
    def forward(ctx, pred, target, kind, delta, loss_weight):
        lib = _C.load()
        pred = _C.require_device(pred, "pred").contiguous()
        target = _C.require_device(target, "target").contiguous()
        if pred.shape != target.shape:
            raise _C.NeosrAmdError(f"loss: shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
        out = torch.empty((), device=pred.device, dtype=torch.float32)
        ws = torch.empty(1024, device=pred.device, dtype=torch.float32)
        _C.check(lib.neosr_pointwise_loss_fwd(pred.data_ptr(), target.data_ptr(), pred.numel(), kind, delta,
                                              loss_weight, out.data_ptr(), ws.data_ptr(), _C.stream_ptr()),
                 "neosr_pointwise_loss_fwd")
        ctx.save_for_backward(pred, target)
        ctx.cfg = (kind, delta, loss_weight)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        pred, target = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        gp = torch.empty_like(pred)
        _C.check(lib.neosr_pointwise_loss_bwd(pred.data_ptr(), target.data_ptr(), g.data_ptr(), pred.numel(), *ctx.cfg,
                                              gp.data_ptr(), _C.stream_ptr()), "neosr_pointwise_loss_bwd")
        return gp, (-gp if ctx.needs_input_grad[1] else None), None, None, None


def _check_reduction(reduction: str, who: str) -> None:
    if reduction not in _reduction_modes:
        raise ValueError(f"Unsupported reduction mode: {reduction}. Supported ones are: {_reduction_modes}")
    if reduction == "none":
        raise NotImplementedError(f"neosr_amd {who}: reduction='none' has no HIP kernel (the training loop needs a scalar)")


@LOSS_REGISTRY.register()
class MSELoss(nn.Module):
    """MSE (L2) loss, `loss_weight * mean((pred - target)^2)` (basic_loss.py:57-86)."""

    def __init__(self, loss_weight: float = 1.0, reduction: str = "mean") -> None:
        super().__init__()
        _check_reduction(reduction, "MSELoss")
        self.loss_weight, self.reduction = loss_weight, reduction

    def forward(self, pred: Tensor, target: Tensor, **kwargs) -> Tensor:  # noqa: ARG002
        w = self.loss_weight * (pred.numel() if self.reduction == "sum" else 1)
        return _PointwiseLoss.apply(pred, target, 1, 1.0, float(w))


@LOSS_REGISTRY.register()
class HuberLoss(nn.Module):
    """Huber loss with threshold `delta` (basic_loss.py:89-127)."""

    def __init__(self, loss_weight: float = 1.0, reduction: str = "mean", delta: float = 1.0) -> None:
        super().__init__()
        _check_reduction(reduction, "HuberLoss")
        self.loss_weight, self.reduction, self.delta = loss_weight, reduction, delta

    def forward(self, pred: Tensor, target: Tensor, **kwargs) -> Tensor:  # noqa: ARG002
        w = self.loss_weight * (pred.numel() if self.reduction == "sum" else 1)
        return _PointwiseLoss.apply(pred, target, 2, float(self.delta), float(w))
