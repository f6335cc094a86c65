"""``gan_loss`` (drop-in for neosr/losses/gan_loss.py:7-82): BCE-with-logits against a constant
real/fake label on a HIP reduction kernel (`neosr_bce_logits_fwd/bwd`); `loss_weight` applies to the
generator only (`is_disc=False`)."""

from __future__ import annotations

from torch import Tensor, nn

from neosr_amd.hip.layers import BceLogits
from neosr_amd.utils.registry import LOSS_REGISTRY


@LOSS_REGISTRY.register()
class gan_loss(nn.Module):
    def __init__(self, gan_type: str = "bce", real_label_val: float = 1.0, fake_label_val: float = 0.0,
                 loss_weight: float = 0.1) -> None:
        super().__init__()
        if gan_type not in {"bce", "mse", "huber"}:
            msg = f"GAN type {gan_type} is not implemented."
            raise NotImplementedError(msg)
        if gan_type != "bce":
            raise NotImplementedError(f"gan_type '{gan_type}': only 'bce' has a HIP kernel so far")
        self.gan_type, self.loss_weight = gan_type, loss_weight
        self.real_label_val, self.fake_label_val = real_label_val, fake_label_val
        self.last_mean: Tensor | None = None

    def forward(self, net_output: Tensor, target_is_real: bool, is_disc: bool = False) -> Tensor:
        target = self.real_label_val if target_is_real else self.fake_label_val
        loss, mean = BceLogits.apply(net_output, float(target), 1.0 if is_disc else float(self.loss_weight))
        self.last_mean = mean  # mean(net_output): `out_d_real` / `out_d_fake` for free
        return loss
