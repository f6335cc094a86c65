from neosr_amd.optimizers.adamw import AdamW

__all__ = ["AdamW"]
