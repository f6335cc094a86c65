"""Pixel losses (drop-in for neosr/losses/basic_loss.py).

``L1Loss`` with reduction="mean" — the configuration on the benchmarked path — runs on the HIP
reduction kernels (`neosr_l1_loss_fwd/bwd`: fixed-order two-stage sum, run-to-run deterministic).
Other reductions are rejected loudly rather than silently computed elsewhere.
"""

from __future__ import annotations

from torch import Tensor, nn

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
        if reduction != "mean":
            msg = "neosr_amd L1Loss: only reduction='mean' has a HIP kernel (the hot-path setting)"
            raise NotImplementedError(msg)
        self.loss_weight = loss_weight
        self.reduction = reduction

    def forward(self, pred: Tensor, target: Tensor, **kwargs) -> Tensor:  # noqa: ARG002
        return L1LossFunction.apply(pred, target, self.loss_weight)
