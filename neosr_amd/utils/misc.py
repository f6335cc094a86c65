"""Small host utilities (seeding, logging) used on the training path."""

from __future__ import annotations

import logging
import random

import torch

from neosr_amd.utils.dist_util import get_dist_info

_initialized: set[str] = set()


def set_random_seed(seed: int) -> None:
    """neosr/utils/misc.py:43-46: python `random` + torch global generator."""
    random.seed(seed)
    torch.manual_seed(seed)


def get_root_logger(name: str = "neosr", level: int = logging.INFO) -> logging.Logger:
    """Rank-0 stdout logger (neosr/utils/logger.py:158-207; other ranks only log errors)."""
    logger = logging.getLogger(name)
    if name in _initialized:
        return logger
    handler = logging.StreamHandler()
    handler.setFormatter(logging.Formatter("%(asctime)s %(levelname)s: %(message)s"))
    logger.addHandler(handler)
    logger.propagate = False
    logger.setLevel(level if get_dist_info()[0] == 0 else logging.ERROR)
    _initialized.add(name)
    return logger


class tc:
    """terminal colours used in the reference's messages (neosr/utils/misc.py)."""

    red = "\033[31m"
    light_blue = "\033[94m"
    end = "\033[0m"
