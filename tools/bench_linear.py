#!/usr/bin/env python
"""nn.Linear forward GEMMs of a transformer block with their real epilogues (bias; + residual + DropPath scale; GELU with
the pre-activation kept) on the swinir_medium / hat_l token counts.  usage: python tools/bench_linear.py [M]"""
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


rs = torch.ones(M // 4096, device=dev)
for (N, K, name, kw) in [(540, 180, "qkv  (bias)", {}), (180, 180, "proj (bias+res+rs)", {"res": 1, "rs": 1}),
                         (360, 180, "fc1  (bias+gelu+aux)", {"gelu": 1}), (180, 360, "fc2  (bias+res+rs)", {"res": 1, "rs": 1})]:
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    out, res, aux = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    t0 = timeit(lambda: tr.gemm(_C.GEMM_NT, x, w, M, N, K, out=out))
    t1 = timeit(lambda: tr.gemm(_C.GEMM_NT, x, w, M, N, K, out=out, bias=b, res=res if kw.get("res") else None,
                                row_scale=rs if kw.get("rs") else None, rows_per_scale=4096 if kw.get("rs") else 0,
                                gelu=bool(kw.get("gelu")), aux_out=aux if kw.get("gelu") else None))
    print(f"{name:22s} M={M} N={N} K={K}: bare {t0:6.1f} us {fl / t0 / 1e6:5.1f} TF | with epilogue {t1:6.1f} us {fl / t1 / 1e6:5.1f} TF")
