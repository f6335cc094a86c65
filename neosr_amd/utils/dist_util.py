"""Process-group bootstrap: one process per MI355X, RCCL over xGMI.

Mirrors neosr/utils/dist_util.py:12-84 (``init_dist``, ``get_dist_info``, ``master_only``) with
the same env-var contract (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK).  On
PyTorch-ROCm ``backend="nccl"`` *is* RCCL.  Unlike the reference, ``init_dist`` is idempotent:
the reference re-enters it from every import-time ``parse_options``.
"""

from __future__ import annotations

import functools
import os
from collections.abc import Callable

import torch
import torch.distributed as dist


def _default_backend() -> str:
    return "nccl" if torch.cuda.is_available() else "gloo"


def init_dist(launcher: str, backend: str | None = None, **kwargs) -> None:
    if dist.is_available() and dist.is_initialized():
        return
    backend = backend or _default_backend()
    if launcher == "pytorch":
        _init_dist_pytorch(backend, **kwargs)
    else:   # ("slurm" in the reference is multi-node glue: one node of 8 MI355X over xGMI is this path's scope)
        msg = f"Invalid launcher type: {launcher} (neosr_amd runs one process per GPU of ONE node: use 'pytorch')"
        raise ValueError(msg)


def _bind_device(index: int) -> None:
    if torch.cuda.is_available():
        torch.cuda.set_device(index % torch.cuda.device_count())


def _init_dist_pytorch(backend: str, **kwargs) -> None:
    rank = int(os.environ["RANK"])
    _bind_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group(backend=backend, **kwargs)


def get_dist_info() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def master_only(func: Callable) -> Callable:
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        if get_dist_info()[0] == 0:
            return func(*args, **kwargs)
        return None

    return wrapper
