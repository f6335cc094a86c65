#!/usr/bin/env python
"""Golden fixture for `train.eco` (image.py:393-418,441-448), produced by RUNNING THE REFERENCE on CPU (build container only):
    python tests/golden/gen_golden_eco.py

  step_eco.npz   5 iterations of the reference `image` model (tiny compact, L1, AdamW, clip, EMA) with eco = true,
                 eco_iters = 4, eco_init = 2, sigmoid schedule: iteration 1 is a plain step (before eco_init, no pretrain),
                 2-4 go through eco_strategy (no-grad prediction -> GT / LQ centroids -> prediction from the LQ centroid),
                 5 is plain again (past eco_iters); batches, per-iteration loss and output, final weights and EMA
"""

from __future__ import annotations

import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import REF, TOML_TMPL, install_reference, np_state, save  # noqa: E402

NET = 'type = "compact"\nnum_feat = 16\nnum_conv = 3'
ECO = 'grad_clip = true\neco = true\neco_iters = 4\neco_init = 2\neco_schedule = "sigmoid"'


def main():
    toml = TOML_TMPL.format(arch="eco", net=NET).replace("grad_clip = true", ECO)
    tmp = Path(tempfile.mkdtemp()) / "golden_eco.toml"
    tmp.write_text(toml)
    (HERE / "golden_eco.toml").write_text(toml)
    install_reference(str(tmp))
    from neosr.models import build_model
    from neosr.utils.options import parse_options

    opt, _ = parse_options(str(REF), is_train=True)
    torch.manual_seed(1024)
    random.seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    arrays = {f"init/{k}": v for k, v in np_state(model.get_bare_model(model.net_g).state_dict()).items()}
    gen = torch.Generator().manual_seed(199)
    logs = []
    for it in range(1, 6):
        lq = torch.rand(2, 3, 16, 16, generator=gen)
        gt = torch.rand(2, 3, 64, 64, generator=gen)
        arrays[f"lq{it}"], arrays[f"gt{it}"] = lq.numpy(), gt.numpy()
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it)
        log = model.get_current_log()
        logs.append([log["l_g_pix"], log["l_g_total"]])
        arrays[f"out{it}"] = model.output.detach().numpy().copy()
        arrays[f"gt_used{it}"] = model.gt.detach().numpy().copy()
    arrays["log"] = np.asarray(logs, dtype=np.float64)
    for k, v in np_state(model.net_g.state_dict()).items():
        arrays[f"final/{k}"] = v
    for k, v in np_state(model.net_g_ema.state_dict()).items():
        arrays[f"ema/{k}"] = v
    save("step_eco.npz", **arrays)


if __name__ == "__main__":
    main()
