"""``gan_loss`` (drop-in for neosr/losses/gan_loss.py:7-82): BCE-with-logits (`neosr_bce_logits_fwd/bwd`),
MSE or Huber (`neosr_pointwise_loss_*`) against a constant real/fake label; `loss_weight` applies to the
generator only (`is_disc=False`)."""

from __future__ import annotations

import torch
from torch import Tensor, nn

from neosr_amd import _C
from neosr_amd.hip.layers import BceLogits
from neosr_amd.losses.basic_loss import _PointwiseLoss
from neosr_amd.utils.registry import LOSS_REGISTRY


@LOSS_REGISTRY.register()
class gan_loss(nn.Module):
    def __init__(self, gan_type: str = "bce", real_label_val: float = 1.0, fake_label_val: float = 0.0,
                 loss_weight: float = 0.1) -> None:
        super().__init__()
        if gan_type not in {"bce", "mse", "huber"}:
            msg = f"GAN type {gan_type} is not implemented."
            raise NotImplementedError(msg)
        self.gan_type, self.loss_weight = gan_type, loss_weight
        self.real_label_val, self.fake_label_val = real_label_val, fake_label_val
        self.last_mean: Tensor | None = None

    def forward(self, net_output: Tensor, target_is_real: bool, is_disc: bool = False) -> Tensor:
        target = self.real_label_val if target_is_real else self.fake_label_val
        weight = 1.0 if is_disc else float(self.loss_weight)
        if self.gan_type == "bce":
            loss, mean = BceLogits.apply(net_output, float(target), weight)
            self.last_mean = mean  # mean(net_output): `out_d_real` / `out_d_fake` for free
            return loss
        # nn.MSELoss / nn.HuberLoss (delta 1) against net_output.new_ones(...) * target (gan_loss.py:45-57)
        label = torch.full_like(net_output, float(target))
        loss = _PointwiseLoss.apply(net_output, label, 1 if self.gan_type == "mse" else 2, 1.0, weight)
        with torch.no_grad():  # logging value mean(net_output) on the HIP column-sum kernel
            lib, x = _C.load(), net_output.detach().contiguous()
            mean = torch.empty(1, device=x.device, dtype=torch.float32)
            ws = torch.empty(128, device=x.device, dtype=torch.float32)
            _C.check(lib.neosr_batched_colsum(x.data_ptr(), None, mean.data_ptr(), ws.data_ptr(), 1, x.numel(), 1,
                                              1.0 / x.numel(), _C.stream_ptr()), "neosr_batched_colsum")
        self.last_mean = mean.reshape(())
        return loss
