#!/usr/bin/env python
"""8x8 window attention forward / backward timing on the swinir_medium shape (GPU box only)."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from neosr_amd.hip import transformer as tr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C, heads = 180, 6
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = B * 64 * 64
qkv = torch.randn(B, 64, 64, 3 * C, device=dev, requires_grad=True)
tab = torch.randn(225, heads, device=dev, requires_grad=True)
for shift in (0, 4):
    t = timeit(lambda: tr.window_attention(qkv.detach(), tab.detach(), heads, 8, shift, 30 ** -0.5))
    o = tr.window_attention(qkv, tab, heads, 8, shift, 30 ** -0.5)
    go = torch.randn_like(o)
    tb = timeit(lambda: torch.autograd.grad(o, (qkv,), go, retain_graph=True))
    fl = 4.0 * M * 64 * C
    print(f"wattn shift {shift}: fwd {t:7.1f} us ({fl / t / 1e6:5.1f} TF)  bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:5.1f} TF)")
