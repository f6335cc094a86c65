"""neosr_amd — MI355X-native (gfx950 / CDNA4) implementation of neosr's training hot path.

Plugin surface kept from muslll/neosr: ``ARCH_REGISTRY`` / ``LOSS_REGISTRY`` / ``MODEL_REGISTRY``,
``build_network`` / ``build_loss`` / ``build_model``, ``parse_options`` (TOML).  Everything that
touches pixels or weights runs in hand-written HIP kernels behind the C ABI of
``include/neosr_amd.h`` (``neosr_amd/lib/libneosr_amd.so``).
"""

from neosr_amd.utils.registry import (  # noqa: F401
    ARCH_REGISTRY,
    DATASET_REGISTRY,
    LOSS_REGISTRY,
    METRIC_REGISTRY,
    MODEL_REGISTRY,
)

__version__ = "0.1.0"
