"""CPU: the float64 / numpy checks DESIGN.md cites for the Winograd forms (tools/micro/): the F(4x4-tile) weight-gradient
identity conv3x3_wgrad_wino4_kernel implements, and the rounding of F(2x2,3x3) / F(4x4,3x3) against a float64 convolution
(the figures quoted in DESIGN.md section 3: ~3e-7 and ~6e-6 of the output scale)."""

from __future__ import annotations

import re
import runpy

from tests.conftest import ROOT


def test_wgrad_f43_identity_holds_in_float64(capsys):
    runpy.run_path(str(ROOT / "tools" / "micro" / "wgrad43_check.py"), run_name="__main__")
    assert "ok" in capsys.readouterr().out


def test_winograd_rounding_is_what_design_md_quotes(capsys):
    runpy.run_path(str(ROOT / "tools" / "micro" / "wino43_err.py"), run_name="__main__")
    out = capsys.readouterr().out
    err = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^(.+?)\s+max \|err\| / max \|y\| = ([0-9.e+-]+)", out, re.M)}
    assert err["direct fp32"] < 1e-6 and err["F(2x2,3x3)"] < 2e-6
    assert 1e-6 < err["F(4x4,3x3)"] < 2e-5     # ~10x the others, far inside north_star's 1e-3
