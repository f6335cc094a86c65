#!/usr/bin/env python
"""Does hipGraph replay of esrgan forward + L1 + backward beat eager launches? (GPU box only)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from neosr_amd.archs import build_network
from neosr_amd.hip.nets import L1LossFunction, flatten_parameters_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
net = build_network({"type": "esrgan", "scale": 4}).cuda().train()
flatten_parameters_(net)
lq, gt = torch.rand(B, 3, 64, 64, device="cuda"), torch.rand(B, 3, 256, 256, device="cuda")


def step():
    net.zero_grad(set_to_none=True)
    out = net(lq)
    loss = L1LossFunction.apply(out, gt, 1.0)
    loss.backward()


def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    step()
print(f"eager fwd+bwd: {timeit(step):.2f} ms")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
net.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    out = net(lq)
    loss = L1LossFunction.apply(out, gt, 1.0)
    loss.backward()
g.replay()
print(f"graph fwd+bwd: {timeit(g.replay):.2f} ms")
