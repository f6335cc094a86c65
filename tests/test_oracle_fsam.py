"""CPU: the Friendly-SAM restatement in oracle/ against fixtures produced by running the reference
(tests/golden/gen_golden_fsam.py)."""

from __future__ import annotations

import numpy as np
import torch

from oracle import neosr_oracle as orc
from tests.conftest import group, load_golden, rel_err


def test_fsam_steps_vs_reference_fixture():
    G = load_golden("fsam.npz")
    ps = [torch.from_numpy(G[f"p0/{i}"].copy()) for i in range(2)]
    sam = orc.FSAM(ps, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.01)
    for step in range(1, 5):
        sam.first_step([torch.from_numpy(G[f"g{step}/{i}"].copy()) for i in range(2)])
        for i in range(2):
            assert rel_err(ps[i], torch.from_numpy(G[f"pert{step}/{i}"])) < 1e-6
        sam.second_step([torch.from_numpy(G[f"h{step}/{i}"].copy()) for i in range(2)])
        for i in range(2):
            assert rel_err(ps[i], torch.from_numpy(G[f"p{step}/{i}"])) < 1e-6
            assert rel_err(sam.momentum[i], torch.from_numpy(G[f"mom{step}/{i}"])) < 1e-6


def test_fsam_image_trajectory_vs_reference_fixture():
    G = load_golden("step_fsam.npz")
    tr = orc.FsamImageTrainer(lambda P, x: orc.rrdbnet_forward(P, x, 4), group(G, "init"), lr=2e-4, betas=(0.9, 0.99),
                              weight_decay=0.01, sam_init=3)
    for it in range(1, 6):
        tr.feed_data(torch.from_numpy(G[f"lq{it}"]), torch.from_numpy(G[f"gt{it}"]))
        tr.optimize_parameters(it)
        assert abs(tr.log["l_g_pix"] - G["log"][it - 1, 0]) < 2e-5 * abs(G["log"][it - 1, 0])
        assert rel_err(tr.output, torch.from_numpy(G[f"out{it}"])) < 2e-5
    for k, v in group(G, "final").items():
        assert rel_err(tr.P[k].detach(), v) < 2e-5, k
    ema = dict(zip(tr.names, tr.ema))
    for k, v in group(G, "ema").items():
        k = k.removeprefix("module.")
        if k in ema:
            assert rel_err(ema[k], v) < 2e-5, k
    for k, v in group(G, "momentum").items():
        assert rel_err(tr.sam.momentum[tr.names.index(k)], v) < 1e-4, k
