from neosr_amd.optimizers.adamw import AdamW
from neosr_amd.optimizers.adan_sf import adan_sf

__all__ = ["AdamW", "adan_sf"]
