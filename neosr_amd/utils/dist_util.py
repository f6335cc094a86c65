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
    elif launcher == "slurm":
        _init_dist_slurm(backend, **kwargs)
    else:
        msg = f"Invalid launcher type: {launcher}"
        raise ValueError(msg)


def _bind_device(index: int) -> None:
    if torch.cuda.is_available():
        torch.cuda.set_device(index % torch.cuda.device_count())


def _init_dist_pytorch(backend: str, **kwargs) -> None:
    rank = int(os.environ["RANK"])
    _bind_device(int(os.environ.get("LOCAL_RANK", rank)))
    dist.init_process_group(backend=backend, **kwargs)


def _slurm_rendezvous(env: dict[str, str], first_host: str, ngpu: int, port: int | None) -> dict[str, str]:
    """The env-var contract of `torch.distributed` derived from a SLURM allocation (neosr/utils/dist_util.py:37-69):
    SLURM_PROCID -> RANK, SLURM_NTASKS -> WORLD_SIZE, the first host of SLURM_NODELIST -> MASTER_ADDR; an explicit `port`
    wins over an exported MASTER_PORT, which wins over 29500.  Pure function of its inputs (tested on CPU)."""
    proc_id = int(env["SLURM_PROCID"])
    out = {
        "RANK": str(proc_id),
        "WORLD_SIZE": str(int(env["SLURM_NTASKS"])),
        "LOCAL_RANK": str(proc_id % max(ngpu, 1)),
        "MASTER_ADDR": first_host,
        "MASTER_PORT": str(port) if port is not None else env.get("MASTER_PORT", "29500"),
    }
    return out


def _init_dist_slurm(backend: str, port: int | None = None) -> None:
    import subprocess

    node_list = os.environ["SLURM_NODELIST"]
    first_host = subprocess.getoutput(f"scontrol show hostname {node_list} | head -n1")
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 1
    os.environ.update(_slurm_rendezvous(dict(os.environ), first_host, ngpu, port))
    _bind_device(int(os.environ["SLURM_PROCID"]))
    dist.init_process_group(backend=backend)


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
