"""debug: sensitivity of the hat_l fixture comparison (tests/test_hip_cfgs.py) to last-bit changes of the conv weight images"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.conftest import load_golden, rel_err
from neosr_amd.archs import hat_arch as A
from neosr_amd import _C
T = torch.from_numpy
fix = load_golden("cfg4_hat_l.npz")
seed = int(fix["seed"])
lib = _C.load()
for mode in (2,):
    lib.neosr_set_winograd(mode)
    torch.manual_seed(seed)
    net = A.hat_l(upscale=4, drop_path_rate=0.0)
    sgen = torch.Generator().manual_seed(9000 + seed)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(torch.randn(p.shape, generator=sgen) * 0.02)
    keys = [str(k) for k in fix["p/keys"]]
    net = net.cuda().train()
    x = T(fix["x"]).cuda().requires_grad_(True)
    from neosr_amd.hip.transformer import ChannelGate
    ChannelGate.trace = []
    y = net(x)
    tr, ChannelGate.trace = ChannelGate.trace, None
    pre = torch.cat([(p.detach().double().cpu() @ w.detach().double().cpu().reshape(w.shape[0], -1).T + b.detach().double().cpu()).flatten() for p, w, b in tr]).numpy()
    if "ca/pre" in fix:
        d = np.abs(pre - fix["ca/pre"])
        print(f"   channel-attention ReLU inputs: max |ours - reference| {d.max():.2e} (unit {d.argmax()}), reference min |input| {np.abs(fix['ca/pre']).min():.2e}")
    y.backward(T(fix["r"]).cuda())
    P = dict(net.named_parameters())
    l2 = np.array([float(P[k].grad.double().norm()) for k in keys])
    e_l2 = np.max(np.abs(l2 - fix["g/l2"]) / (fix["g/l2"] + 1e-12))
    s = np.array([float(P[k].grad.double().sum()) for k in keys])
    e_s = np.max(np.abs(s - fix["g/sum"]) / (fix["g/abs"] + 1e-12))
    full = max(rel_err(P[k[len("gfull/"):]].grad, T(fix[k])) for k in fix if k.startswith("gfull/"))
    gx = x.grad.detach().cpu().double().flatten(); rx = T(fix["gx"]).double().flatten()
    cos = float((gx @ rx) / (gx.norm() * rx.norm()))
    rel = np.abs(l2 - fix["g/l2"]) / (fix["g/l2"] + 1e-12)
    top = np.argsort(-rel)[:8]
    print("   worst parameter gradients:", [(keys[i], float(f"{rel[i]:.2e}")) for i in top])
    print(f"winograd mode {mode}: y {rel_err(y, T(fix['y'])):.2e}  dx {rel_err(x.grad, T(fix['gx'])):.2e} (cos {cos:.8f})  "
          f"param-grad l2 {e_l2:.2e} sum {e_s:.2e} full {full:.2e}")
