"""Helpers shared by the arch plugins."""

from __future__ import annotations

import torch
from torch import nn

from neosr_amd.hip.nets import flatten_parameters_, parameters_of
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
        return parameters_of(self)


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


class DropPathBank:
    """Per-sample DropPath scales (Bernoulli(keep) / keep, neosr/archs/arch_util.py:118-133) for ALL stochastic-depth
    sites of one generator forward, drawn in two launches instead of two per site (a transformer generator has 70-140
    sites).  The first train-mode forward draws site by site and records the sequence of drop probabilities; from then
    on `begin()` draws the whole (sites, batch) matrix at once and `scale()` hands out its rows in call order.  A call
    that does not match the recorded sequence (another batch size, changed probabilities) falls back to a draw of its
    own."""

    def __init__(self) -> None:
        self.probs: list[float] = []
        self.recorded = False
        self._keep = None      # (sites, 1) tensor
        self._rows = None
        self._pos = 0
        self._b = -1

    def begin(self, training: bool, b: int, device) -> None:
        self._rows, self._pos, self._b = None, 0, b
        if not training:
            return
        if not self.recorded:
            self.probs = []
            return
        if not self.probs:
            return
        if self._keep is None or self._keep.device != device:
            self._keep = torch.tensor([1.0 - p for p in self.probs], dtype=torch.float32, device=device).view(-1, 1)
        rows = torch.bernoulli(self._keep.expand(-1, b))
        # (a site with drop_prob == 1 keeps nothing: its scale stays 0 like the per-site path / the reference's
        # `keep_prob > 0` guard, instead of 0 / 0)
        self._rows = rows.div_(self._keep.clamp_min(torch.finfo(torch.float32).tiny))

    def end(self, training: bool) -> None:
        if training and not self.recorded:
            self.recorded = True
        self._rows = None

    def scale(self, drop_prob: float, training: bool, b: int, device):
        if drop_prob == 0.0 or not training:
            return None
        if self._rows is not None and self._pos < len(self.probs) and self.probs[self._pos] == drop_prob and b == self._b:
            self._pos += 1
            return self._rows[self._pos - 1]
        if not self.recorded:
            self.probs.append(drop_prob)
        keep = 1.0 - drop_prob
        rs = torch.empty(b, device=device, dtype=torch.float32).bernoulli_(keep)
        if keep > 0.0:
            rs.div_(keep)
        return rs


_ACTIVE_BANK: DropPathBank | None = None


class drop_path_bank:
    """`with drop_path_bank(net_bank, training, batch, device):` around the block loop of a generator forward."""

    def __init__(self, bank: DropPathBank, training: bool, b: int, device) -> None:
        self.bank, self.training, self.b, self.device = bank, training, b, device

    def __enter__(self):
        global _ACTIVE_BANK
        self.bank.begin(self.training, self.b, self.device)
        _ACTIVE_BANK = self.bank
        return self.bank

    def __exit__(self, *exc):
        global _ACTIVE_BANK
        _ACTIVE_BANK = None
        if exc[0] is None:
            self.bank.end(self.training)
        return False


def drop_scale(drop_prob: float, training: bool, b: int, device):
    """DropPath scale of one site: a row of the active bank, or (no bank: a block called on its own) its own draw."""
    bank = _ACTIVE_BANK if _ACTIVE_BANK is not None else DropPathBank()
    return bank.scale(drop_prob, training, b, device)
