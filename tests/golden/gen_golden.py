#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference (/root/reference) on CPU.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/gen_golden.py

What it does: puts tiny import shims (tests/golden/_shims: tomllib->tomli, constant-only cv2,
a torchvision skeleton, empty lmdb/pywt) and /root/reference on sys.path, redirects "cuda" to
the CPU, then runs the reference's own modules (`esrgan`, `compact`, `L1Loss`, the `image`
model's `feed_data`/`optimize_parameters`) on seeded inputs and stores inputs + outputs as
small .npz files.  The fixtures are data only; no reference source is copied.
"""

from __future__ import annotations

import json
import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")

TOML_TMPL = """
name = "golden_{arch}"
model_type = "image"
scale = 4
manual_seed = 1024

[datasets.train]
type = "paired"
dataroot_gt = "/tmp/none_gt"
dataroot_lq = "/tmp/none_lq"
patch_size = 16
batch_size = 2

[path]

[network_g]
{net}

[train]
ema = 0.999
grad_clip = true

[train.optim_g]
type = "adamw"
lr = 1e-3
betas = [ 0.9, 0.99 ]
weight_decay = 0.01

[train.pixel_opt]
type = "L1Loss"
loss_weight = 1.0

[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


def install_reference(cfg_path: str) -> None:
    sys.path[:0] = [str(HERE / "_shims"), str(REF)]
    sys.argv = ["gen_golden", "-opt", cfg_path]
    # cuda -> cpu redirect
    _to = torch.Tensor.to
    _mto = torch.nn.Module.to

    def fix(a):
        if isinstance(a, str) and a.startswith("cuda"):
            return "cpu"
        if isinstance(a, torch.device) and a.type == "cuda":
            return torch.device("cpu")
        return a

    def tensor_to(self, *args, **kw):
        args = tuple(fix(a) for a in args)
        kw = {k: fix(v) for k, v in kw.items()}
        kw.pop("non_blocking", None)
        return _to(self, *args, **kw)

    def module_to(self, *args, **kw):
        args = tuple(fix(a) for a in args)
        kw = {k: fix(v) for k, v in kw.items()}
        return _mto(self, *args, **kw)

    def wrap_factory(fn):
        def inner(*args, **kw):
            if "device" in kw:
                kw["device"] = fix(kw["device"])
            return fn(*args, **kw)

        return inner

    for fname in ("tensor", "zeros", "ones", "empty", "rand", "randn", "full", "arange",
                  "zeros_like", "ones_like", "empty_like", "as_tensor", "randperm"):
        setattr(torch, fname, wrap_factory(getattr(torch, fname)))
    torch.Tensor.to = tensor_to
    torch.nn.Module.to = module_to
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.manual_seed = lambda *a, **k: None
    torch.cuda.manual_seed_all = lambda *a, **k: None


def np_state(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def save(name: str, **arrays) -> None:
    path = HERE / name
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({path.stat().st_size / 1024:.1f} KiB)")


def fwd_bwd_fixture(name, net, x, gt, extra=None):
    net.train()
    x = x.clone()
    y = net(x)
    loss = torch.nn.functional.l1_loss(y, gt)
    loss.backward()
    arrays = {"x": x.numpy(), "gt": gt.numpy(), "y": y.detach().numpy(),
              "loss": np.float32(loss.item())}
    for k, v in net.state_dict().items():
        arrays[f"param/{k}"] = v.detach().numpy().copy()
    for k, p in net.named_parameters():
        arrays[f"grad/{k}"] = p.grad.detach().numpy().copy()
    arrays.update(extra or {})
    save(name, **arrays)


def gen_archs():
    from neosr.archs.compact_arch import compact
    from neosr.archs.esrgan_arch import esrgan

    torch.manual_seed(7)
    net = esrgan(num_in_ch=3, num_out_ch=3, scale=4, num_feat=16, num_block=2, num_grow_ch=8)
    with torch.no_grad():  # non-zero biases so the bias path is exercised
        for n, p in net.named_parameters():
            if n.endswith("bias"):
                p.uniform_(-0.1, 0.1)
    x = torch.rand(2, 3, 12, 20)
    gt = torch.rand(2, 3, 48, 80)
    fwd_bwd_fixture("esrgan_small.npz", net, x, gt)

    for act in ("prelu", "leakyrelu", "relu"):
        torch.manual_seed(11)
        net = compact(num_in_ch=3, num_out_ch=3, num_feat=16, num_conv=3, upscale=4, act_type=act)
        if act == "prelu":  # spread the slopes (default 0.25 everywhere), include a negative one
            with torch.no_grad():
                for n, p in net.named_parameters():
                    if p.dim() == 1 and "weight" in n:
                        p.uniform_(-0.2, 0.5)
        x = torch.rand(2, 3, 10, 14)
        gt = torch.rand(2, 3, 40, 56)
        fwd_bwd_fixture(f"compact_small_{act}.npz", net, x, gt)


def gen_index():
    arrays = {}
    for r in (2, 4):
        c = 3
        x = torch.arange(2 * c * r * r * 5 * 7, dtype=torch.float32).reshape(2, c * r * r, 5, 7)
        arrays[f"ps{r}_in"] = x.numpy()
        arrays[f"ps{r}_out"] = torch.nn.PixelShuffle(r)(x).numpy()
    from neosr.archs.esrgan_arch import pixel_unshuffle

    for r in (2, 4):
        x = torch.arange(2 * 3 * 8 * 12, dtype=torch.float32).reshape(2, 3, 8, 12)
        arrays[f"pu{r}_in"] = x.numpy()
        arrays[f"pu{r}_out"] = pixel_unshuffle(x, r).numpy()
    save("index.npz", **arrays)


def gen_loss():
    from neosr.losses.basic_loss import L1Loss

    torch.manual_seed(3)
    a = torch.rand(2, 3, 17, 23, requires_grad=True)
    b = torch.rand(2, 3, 17, 23)
    out = L1Loss(loss_weight=0.7)(a, b)
    out.backward()
    save("l1loss.npz", pred=a.detach().numpy(), target=b.numpy(), loss=np.float32(out.item()),
         grad=a.grad.numpy())


def gen_step(arch: str, opt, n_iter: int = 3):
    """`n_iter` iterations of the reference image model: feed_data + optimize_parameters."""
    from neosr.models import build_model

    torch.manual_seed(1024)
    random.seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    init = np_state(model.get_bare_model(model.net_g).state_dict())
    gen = torch.Generator().manual_seed(99)
    arrays = {f"init/{k}": v for k, v in init.items()}
    logs = []
    for it in range(1, n_iter + 1):
        lq = torch.rand(2, 3, 16, 16, generator=gen)
        gt = torch.rand(2, 3, 64, 64, generator=gen)
        arrays[f"lq{it}"] = lq.numpy()
        arrays[f"gt{it}"] = gt.numpy()
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it)
        log = model.get_current_log()
        logs.append([log["l_g_pix"], log["l_g_total"]])
        arrays[f"out{it}"] = model.output.detach().numpy().copy()
    arrays["log"] = np.asarray(logs, dtype=np.float64)
    for k, v in np_state(model.net_g.state_dict()).items():
        arrays[f"final/{k}"] = v
    for k, v in np_state(model.net_g_ema.state_dict()).items():
        arrays[f"ema/{k}"] = v
    st = model.optimizer_g.state_dict()["state"]
    names = [n for n, _ in model.net_g.named_parameters()]
    for i in (0, len(names) - 1):
        arrays[f"adam_exp_avg/{names[i]}"] = st[i]["exp_avg"].numpy().copy()
        arrays[f"adam_exp_avg_sq/{names[i]}"] = st[i]["exp_avg_sq"].numpy().copy()
    save(f"step_{arch}.npz", **arrays)


def dump_opt(opt, name):
    def conv(o):
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, Path):
            return "<path>/" + o.name
        if isinstance(o, (list, tuple)):
            return [conv(v) for v in o]
        return o

    (HERE / name).write_text(json.dumps(conv(opt), indent=1, sort_keys=True) + "\n")
    print("wrote", HERE / name)


def main():
    nets = {
        "compact": 'type = "compact"\nnum_feat = 16\nnum_conv = 3',
        "esrgan": 'type = "esrgan"\nnum_feat = 16\nnum_block = 2\nnum_grow_ch = 8',
    }
    arch = sys.argv[1] if len(sys.argv) > 1 else "compact"
    tmp = Path(tempfile.mkdtemp()) / f"golden_{arch}.toml"
    tmp.write_text(TOML_TMPL.format(arch=arch, net=nets[arch]))
    # also keep the TOML next to the fixtures: our own parse_options is tested on the same file
    (HERE / f"golden_{arch}.toml").write_text(tmp.read_text())
    install_reference(str(tmp))
    from neosr.utils.options import parse_options

    opt, _ = parse_options(str(REF), is_train=True)
    dump_opt(opt, f"opt_{arch}.json")
    if arch == "compact":
        gen_archs()
        gen_index()
        gen_loss()
    gen_step(arch, opt)


if __name__ == "__main__":
    main()
