"""CPU-side checks of the drop-in boundary: libneosr_amd.so loads without a GPU and exports every
symbol include/neosr_amd.h declares; the ctypes table mirrors the header."""

from __future__ import annotations

import re

import pytest

from tests.conftest import ROOT


def _header_functions() -> list[str]:
    text = (ROOT / "include" / "neosr_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(neosr_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from neosr_amd import _C

    lib = _C.load()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/neosr_amd.h but not exported"
    assert lib.neosr_abi_version() == 1
    assert b"gfx950" in lib.neosr_build_info()


def test_ctypes_table_matches_header():
    from neosr_amd import _C

    assert sorted(_C.SIGNATURES) == _header_functions()


def test_struct_sizes_match_c_layout():
    """ctypes mirrors of the descriptor structs must have the C sizes (checked against the
    compiler through a tiny probe translation unit)."""
    import subprocess
    import tempfile
    from pathlib import Path

    from neosr_amd import _C
    import ctypes as C

    src = ('#include "neosr_amd.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
           "sizeof(neosr_conv_desc),sizeof(neosr_wgrad_desc),sizeof(neosr_adamw_desc),"
           "sizeof(neosr_rrdbnet_cfg),sizeof(neosr_compact_cfg),sizeof(neosr_tblock_desc),sizeof(neosr_tblock_grads),"
           "sizeof(neosr_dslope_item),sizeof(neosr_gemm_desc),sizeof(neosr_fattn_desc));return 0;}\n")
    with tempfile.TemporaryDirectory() as td:
        c = Path(td) / "probe.c"
        c.write_text(src)
        exe = Path(td) / "probe"
        subprocess.run(["gcc", "-I", str(ROOT / "include"), str(c), "-o", str(exe)], check=True)
        sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    got = [C.sizeof(t) for t in (_C.ConvDesc, _C.WgradDesc, _C.AdamWDesc, _C.RRDBNetCfg, _C.CompactCfg, _C.TBlockDesc,
                                 _C.TBlockGrads, _C.DslopeItem, _C.GemmDesc, _C.FattnDesc)]
    assert got == sizes


def test_workspace_queries_need_no_gpu():
    import ctypes as C

    from neosr_amd import _C

    lib = _C.load()
    cfg = _C.RRDBNetCfg(16, 64, 64, 3, 3, 64, 23, 32, 1)
    nbytes = lib.neosr_rrdbnet_workspace_bytes(C.byref(cfg))
    assert 3 << 30 < nbytes < 16 << 30          # ~69 x 50 MB activations + HR buffers at B=16
    assert lib.neosr_rrdbnet_num_params(C.byref(cfg)) == 702
    ccfg = _C.CompactCfg(2, 64, 64, 3, 3, 64, 16, 4, _C.ACT_PRELU, 1)
    assert lib.neosr_compact_num_params(C.byref(ccfg)) == 53
    assert lib.neosr_compact_workspace_bytes(C.byref(ccfg)) > 0
    bad = _C.RRDBNetCfg(0, 64, 64, 3, 3, 64, 23, 32, 1)
    assert lib.neosr_rrdbnet_workspace_bytes(C.byref(bad)) == -1
    assert b"bad cfg" in lib.neosr_last_error()


def test_product_path_refuses_cpu_tensors():
    """No silent CPU fallback: a CPU tensor is an error, not a slow path."""
    import torch

    from neosr_amd import _C
    from neosr_amd.archs import build_network

    net = build_network({"type": "compact", "num_feat": 8, "num_conv": 1, "upscale": 4})
    with pytest.raises(_C.NeosrAmdError):
        net(torch.rand(1, 3, 8, 8))
