"""GPU: the data-parallel path over RCCL (backend "nccl") with a single-rank group — RCCL refuses two
ranks on one device, so world_size 2 is covered by tests/test_dist_gloo.py on CPU; this one checks that
RCCL initialises here and that an `image` model with `dist = True` (flat-arena all-reduce, 1/world scale in
the optimizer kernel, loss-dict reduce) trains exactly like the non-distributed model."""

from __future__ import annotations

import os
import socket
import subprocess
import sys
import textwrap

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent("""
    import sys, numpy as np, torch
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from neosr_amd.models import build_model
    from neosr_amd.models.base import allreduce_flat_
    from neosr_amd.utils.dist_util import get_dist_info, init_dist
    from neosr_amd.utils.options import parse_options
    from tests.conftest import GOLDEN, group, load_golden

    init_dist("pytorch")                      # default backend on a HIP device: "nccl" = RCCL
    assert dist.get_backend() == "nccl" and get_dist_info() == (0, 1)
    t = torch.arange(1000, device="cuda", dtype=torch.float32)
    allreduce_flat_(t, bucket_bytes=1024)
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
    fix = load_golden("step_esrgan.npz")
    outs = []
    for use_dist in (False, True):
        opt, _ = parse_options({root!r}, True, argv=["-opt", str(GOLDEN / "golden_esrgan.toml")])
        opt["dist"], opt["rank"], opt["world_size"] = use_dist, 0, 1
        model = build_model(opt)
        model.net_g.load_state_dict(group(fix, "init"))
        for it in (1, 2):
            model.feed_data({{"lq": torch.from_numpy(np.array(fix[f"lq{{it}}"])), "gt": torch.from_numpy(np.array(fix[f"gt{{it}}"]))}})
            model.optimize_parameters(it)
        outs.append((model.get_current_log()["l_g_pix"], [p.detach().clone() for p in model.net_g.parameters()]))
    assert outs[0][0] == outs[1][0]
    assert all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    dist.destroy_process_group()
    print("RCCL_SINGLE_RANK_OK")
""")


def test_rccl_single_rank_group_matches_non_distributed(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "rccl_rank.py"
    script.write_text(SCRIPT.format(root=str(ROOT)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
