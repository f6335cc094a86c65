"""Minimal torchvision stand-in (absent in this image): only what neosr imports at module scope."""
from . import models, transforms, utils  # noqa: F401
