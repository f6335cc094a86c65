#!/usr/bin/env python
"""TN (weight-gradient) GEMM timing on the transformer shapes, kernel + reduction separately (GPU box only)."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from neosr_amd import _C
from neosr_amd.hip import transformer as tr

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
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


out = []
for (N, K, name) in [(540, 180, "qkv"), (180, 180, "proj"), (360, 180, "fc1"), (180, 360, "fc2")]:
    x, g = torch.randn(M, K, device=dev), torch.randn(M, N, device=dev)
    gw, gb = tr._wgrad_pair(g, N, K, True)
    t = timeit(lambda: tr.gemm(_C.GEMM_TN, g, x, N, K, M, out=gw, colsum_a=gb))
    out.append(f"{name} {t:6.1f} us {2.0 * M * N * K / t / 1e6:5.1f} TF")
print(" | ".join(out))
