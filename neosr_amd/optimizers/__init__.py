from neosr_amd.optimizers.adamw import AdamW
from neosr_amd.optimizers.adan_sf import adan_sf
from neosr_amd.optimizers.extra import Adam, NAdam, adamw_sf, adamw_win, adan
from neosr_amd.optimizers.fsam import fsam

__all__ = ["Adam", "AdamW", "NAdam", "adamw_sf", "adamw_win", "adan", "adan_sf", "fsam"]
