#!/usr/bin/env python
"""K / N sweep of neosr_gemm at the token count of the transformer configs: where the short-K shapes lose
their time (prologue / epilogue / tile quantisation vs the steady-state loop).  GPU box only.
   python tools/bench_gemm_sweep.py [M]"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from neosr_amd import _C
from neosr_amd.hip import transformer as tr

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = "cuda"


def timeit(fn, n=20):
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


for N in (64, 192, 180, 512, 540, 576):
    for K in (32, 64, 128, 180, 192, 360, 384, 768, 1536):
        x, w, g = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
        fl = 2.0 * M * N * K
        out = torch.empty(M, N, device=dev)
        t = timeit(lambda: tr.gemm(_C.GEMM_NT, x, w, M, N, K, out=out))
        outk = torch.empty(M, K, device=dev)
        t1 = timeit(lambda: tr.gemm(_C.GEMM_NN, g, w, M, K, N, out=outk))
        t2 = timeit(lambda: tr.gemm(_C.GEMM_TN, g, x, N, K, M))
        print(f"M={M} N={N:4d} K={K:5d}: NT {t:7.1f} us {fl / t / 1e6:6.1f} TF | NN(red N) {t1:7.1f} us {fl / t1 / 1e6:6.1f} TF"
              f" | TN {t2:7.1f} us {fl / t2 / 1e6:6.1f} TF", flush=True)
