"""Helpers shared by the arch plugins."""

from __future__ import annotations

import torch
from torch import nn

from neosr_amd.hip.nets import flatten_parameters_
from neosr_amd.utils.options import net_opt  # noqa: F401  (re-export, neosr/archs/arch_util.py:12)


class HipNet(nn.Module):
    """Base for generators whose forward/backward run as one HIP plan.

    Keeps the parameters in a flat HBM arena (re-flattened lazily after ``.to()`` / ``deepcopy``
    / ``load_state_dict`` re-homed them) so the fused optimizer and the gradient all-reduce see
    one contiguous buffer.
    """

    def flat_parameters(self) -> torch.Tensor:
        return flatten_parameters_(self)

    def _plan_params(self) -> list[torch.Tensor]:
        self.flat_parameters()
        return list(self.parameters())


def droppath_ctor_reseed() -> None:
    """Side effect of constructing the reference's DropPath (neosr/archs/arch_util.py:141-146): its
    `net_opt()` call re-parses the TOML, and parse_options re-seeds python `random` and torch with
    `manual_seed + rank` (neosr/utils/options.py:190-208).  Transformer archs call this at the point
    where the reference builds a DropPath so that seeded initialisations match draw for draw."""
    from neosr_amd.utils.misc import set_random_seed
    from neosr_amd.utils.options import global_opt

    opt = global_opt()
    if opt is not None and opt.get("manual_seed") is not None:
        set_random_seed(int(opt["manual_seed"]) + int(opt.get("rank", 0)))


@torch.no_grad()
def default_init_weights(module_list, scale: float = 1, bias_fill: float = 0, **kwargs) -> None:
    """kaiming-normal * scale, constant bias (neosr/archs/esrgan_arch.py:13-38)."""
    if not isinstance(module_list, list):
        module_list = [module_list]
    for module in module_list:
        for m in module.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, **kwargs)
                m.weight.data *= scale
                if m.bias is not None:
                    m.bias.data.fill_(bias_fill)
