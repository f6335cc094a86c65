"""Forward / backward timing of the esrgan plan, one and two launch chains.  usage: python tools/bench_rrdb.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd import _C
from neosr_amd.archs import build_network

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
net = build_network({"type": "esrgan"}).cuda().train()
x = torch.randn(B, 3, 64, 64, device="cuda")
lib = _C.load()


def run(n=5):
    tf = tb = 0.0
    for i in range(n + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        y = net(x)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        y.backward(torch.ones_like(y) * 1e-3)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if i >= 2:
            tf += t1 - t0; tb += t2 - t1
    return tf / n * 1e3, tb / n * 1e3


for ns in (1, 2, 1, 2):
    lib.neosr_set_num_streams(ns)
    f, b = run()
    print(f"streams={ns}: fwd {f:6.2f} ms  bwd {b:6.2f} ms  total {f + b:6.2f}")
