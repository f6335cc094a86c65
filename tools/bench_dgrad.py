#!/usr/bin/env python
"""nn.Linear backward-data GEMMs dX = dY W: the NN kernel on W (N_out, K_in) against the direct-to-LDS NT kernel on a
transposed copy W^T (K_in, N_out), with the real epilogues (GELU' for fc2's data gradient, DropPath scale).
usage: python tools/bench_dgrad.py [M]"""
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


for (No, Ki, name, aux) in [(540, 180, "qkv", 0), (180, 180, "proj", 0), (360, 180, "fc1", 0), (180, 360, "fc2 (GELU')", 1)]:
    g, w = torch.randn(M, No, device=dev), torch.randn(No, Ki, device=dev)
    wt = w.t().contiguous()
    pre = torch.randn(M, Ki, device=dev) if aux else None
    out = torch.empty(M, Ki, device=dev)
    fl = 2.0 * M * No * Ki
    t0 = timeit(lambda: tr.gemm(_C.GEMM_NN, g, w, M, Ki, No, out=out, aux_in=pre))
    a = out.clone()
    t1 = timeit(lambda: tr.gemm(_C.GEMM_NT, g, wt, M, Ki, No, out=out, aux_in=pre))
    err = ((out - a).abs().max() / a.abs().max()).item()
    print(f"{name:12s} M={M} dX[{Ki}] = dY[{No}] W: NN {t0:6.1f} us {fl / t0 / 1e6:5.1f} TF | NT on W^T {t1:6.1f} us {fl / t1 / 1e6:5.1f} TF  (diff {err:.1e})")
