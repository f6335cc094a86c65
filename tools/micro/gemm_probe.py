#!/usr/bin/env python
"""What a Linear GEMM launch costs besides its loop: the NT kernel at N = 576, K = 192 over a range of M —
time ~ fixed cost per launch + slope per 1 024 rows.  GPU box only.  (Re-created in round 4: the round-3 copy lived in the
untracked experiments/ directory.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from neosr_amd import _C
from neosr_amd.hip import transformer as tr

N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (576, 192)
rows = []
for M in (4096, 8192, 16384, 32768, 65536, 131072, 262144):
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        tr.gemm(_C.GEMM_NT, x, w, M, N, K, out=out, bias=b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        tr.gemm(_C.GEMM_NT, x, w, M, N, K, out=out, bias=b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    rows.append((M, us))
    print(f"M {M:7d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF")
(m0, t0), (m1, t1) = rows[2], rows[-1]
slope = (t1 - t0) / (m1 - m0) * 1024
print(f"time ~ {t0 - slope * m0 / 1024:.1f} us + {slope:.2f} us per 1 024 rows")
