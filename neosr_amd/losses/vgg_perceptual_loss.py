"""``vgg_perceptual_loss`` (drop-in for neosr/losses/vgg_perceptual_loss.py:57-242, non-patch path):
`loss_weight * sum_k w_k * chc(f_k(x)/10, f_k(gt)/10)` with the chc criterion at
(lambda=0, clip 0..1) exactly as the reference builds it (:143-144, 232-236).  The /10 is folded
into the loss kernel (`pre = 0.1`); features of `gt` are computed without a graph."""

from __future__ import annotations

import torch
from torch import Tensor, nn

from neosr_amd.archs.vgg_arch import VGGFeatureExtractor
from neosr_amd.hip.layers import ChcLoss
from neosr_amd.hip.nets import L1LossFunction
from neosr_amd.losses.basic_loss import _PointwiseLoss
from neosr_amd.utils.registry import LOSS_REGISTRY


@LOSS_REGISTRY.register()
class vgg_perceptual_loss(nn.Module):
    def __init__(self, layer_weights: dict[str, float] | None = None, vgg_type: str = "vgg19",
                 use_input_norm: bool = True, range_norm: bool = False, loss_weight: float = 1.0,
                 criterion: str = "chc", patchloss: bool = False, ipk: bool = False,
                 patch_weight: float = 1.0, **kwargs) -> None:  # noqa: ARG002
        super().__init__()
        if patchloss is False and ipk is True:
            raise ValueError("Please enable PatchLoss to use IPK.")
        if patchloss:
            raise NotImplementedError("PatchLoss / IPK are off by default and outside the hot path")
        if criterion not in ("l1", "l2", "huber", "chc"):
            raise NotImplementedError(f"{criterion} criterion not supported.")
        self.criterion_type = criterion
        self.loss_weight = loss_weight
        self.layer_weights = layer_weights if layer_weights is not None else {
            "conv1_2": 0.1, "conv2_2": 0.1, "conv3_4": 1.0, "conv4_4": 1.0, "conv5_4": 1.0}
        self.vgg = VGGFeatureExtractor(layer_name_list=list(self.layer_weights.keys()), vgg_type=vgg_type,
                                       use_input_norm=use_input_norm, range_norm=range_norm)

    _gt_pref = None   # (gt tensor, its features, event on the side stream) left by `prefetch_gt`

    def prefetch_gt(self, gt: Tensor, stream: "torch.cuda.Stream") -> None:
        """The target's features do not depend on the generator: computed on `stream` (which must already wait for whatever
        produced `gt`) while the generator's forward runs on the caller's stream; `forward` picks them up when it is handed
        the same tensor.  Same kernels on the same operands: the loss is bit-identical to the serial evaluation."""
        with torch.cuda.stream(stream), torch.no_grad():
            fg = self.vgg.features_nhwc(gt.detach())
            ev = torch.cuda.Event()
            ev.record(stream)
        self._gt_pref = (gt, fg, ev)

    def forward(self, x: Tensor, gt: Tensor) -> Tensor:
        fx = self.vgg.features_nhwc(x)
        pref, self._gt_pref = self._gt_pref, None
        if pref is not None and pref[0] is gt:
            fg = pref[1]
            torch.cuda.current_stream(x.device).wait_event(pref[2])
        else:
            with torch.no_grad():
                fg = self.vgg.features_nhwc(gt.detach())
        total = None
        for k in fx:
            w = float(self.layer_weights[k])
            if self.criterion_type == "chc":
                # chc_loss(loss_lambda=0, clip_min=0, clip_max=1, criterion="huber") on features / 10
                term = ChcLoss.apply(fx[k], fg[k], 0.1, True, 0.0, 1.0, w)
            elif self.criterion_type == "l1":  # criterion(f/10, g/10): the 1/10 folds into the weight
                term = L1LossFunction.apply(fx[k], fg[k], w / 10)
            elif self.criterion_type == "l2":
                term = _PointwiseLoss.apply(fx[k], fg[k], 1, 1.0, w / 100)
            else:  # huber(d/10, delta 1) == huber(d, delta 10) / 100
                term = _PointwiseLoss.apply(fx[k], fg[k], 2, 10.0, w / 100)
            total = term if total is None else total + term
        return total * self.loss_weight
