#!/usr/bin/env python
"""Golden fixture for HAT with window_size 8 (overlapping window 12), produced by RUNNING THE REFERENCE on CPU (build
container only):   python tests/golden/gen_golden_hat_w8.py

  hat_w8.npz   a tiny `hat` (embed 24, depths (2, 2), 2 heads, window 8: shift 4 in the odd blocks, 12x12 overlapping key
               windows in the OCABs) forward + all gradients on 2x3x16x24 — two windows high, three wide, so the shift mask
               has all nine regions and the overlapping windows run over all four image borders
"""

from __future__ import annotations

import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import install_reference, save  # noqa: E402
from gen_golden_hat import TOML  # noqa: E402


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_hat.toml"
    tmp.write_text(TOML)
    install_reference(str(tmp))
    from neosr.archs import hat_arch as HA

    gen = torch.Generator().manual_seed(17)
    A = {}
    for seed in range(261, 361):
        torch.manual_seed(seed)
        net = HA.hat(img_size=16, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=8, compress_ratio=3,
                     squeeze_factor=6, mlp_ratio=2, drop_path_rate=0.0, upsampler="pixelshuffle")
        sgen = torch.Generator().manual_seed(3000 + seed)
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn(p.shape, generator=sgen) * 0.02)
        x = torch.rand(2, 3, 16, 24, generator=sgen).requires_grad_(True)
        closest = [float("inf")]
        hooks = [m.register_forward_pre_hook(lambda _m, a: closest.__setitem__(0, min(closest[0], float(a[0].abs().min()))))
                 for m in net.modules() if isinstance(m, (torch.nn.LeakyReLU, torch.nn.ReLU))]
        y = net(x)
        for hk in hooks:
            hk.remove()
        print(f"tiny hat (window 8): seed {seed} closest (Leaky)ReLU input to zero {closest[0]:.2e}")
        if closest[0] > 2e-7:
            break
    else:
        raise RuntimeError("no well-conditioned draw found")
    r = torch.randn(y.shape, generator=gen)
    (y * r).sum().backward()
    A["x"], A["r"], A["y"], A["gx"] = x.detach().numpy(), r.numpy(), y.detach().numpy(), x.grad.numpy().copy()
    for k, v in net.named_parameters():
        A[f"p/{k}"] = v.detach().numpy().copy()
        A[f"g/{k}"] = v.grad.numpy().copy()
    A["keys"] = np.array(list(net.state_dict().keys()))
    save("hat_w8.npz", **A)


if __name__ == "__main__":
    main()
