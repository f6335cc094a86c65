"""A/B on one box: esrgan B=16 training step with the trunk as chain launches vs per-convolution launches."""
import json, sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neosr_amd import _C
from neosr_amd.archs import build_network

lib = _C.load()
torch.manual_seed(0)
net = build_network({"type": "esrgan", "scale": 4}).cuda().train()
x = torch.rand(16, 3, 64, 64, device="cuda")
gy = torch.randn(16, 3, 256, 256, device="cuda") * 1e-3

def run(n):
    for _ in range(n):
        net.zero_grad(set_to_none=True)
        y = net(x)
        y.backward(gy)

def timed(tag, steps=20):
    run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"mode": tag, "ms_per_fwd_bwd": round(ms, 3)}), flush=True)
    return ms

for rep in range(2):
    lib.neosr_set_conv_chain(0)
    timed("per-layer")
    lib.neosr_set_conv_chain(1)
    lib.neosr_set_conv_chain_sync(1)
    timed("chain")
    lib.neosr_set_conv_chain_sync(0)
    timed("chain-nosync(racy)")
    lib.neosr_set_conv_chain_sync(1)
print("status", lib.neosr_conv_chain_status())
