#!/usr/bin/env python
"""Micro-benchmark of neosr_gemm / window attention / LayerNorm on the SwinIR shapes (GPU box only).
   python tools/bench_gemm.py [B]      (B = batch of 64x64 LR patches, default 8)"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from neosr_amd import _C
from neosr_amd.hip import transformer as tr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = B * 64 * 64
dev = "cuda"


def timeit(fn, n=20):
    for _ in range(10):   # (the first calls of a process run slower: clocks and caches)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


C = 180
for (N, K, name) in [(540, 180, "qkv"), (180, 180, "proj"), (360, 180, "fc1"), (180, 360, "fc2")]:
    x, w, g = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(M, N, device=dev)
    b = torch.randn(N, device=dev)
    fl = 2.0 * M * N * K
    out = torch.empty(M, N, device=dev)
    t = timeit(lambda: tr.gemm(_C.GEMM_NT, x, w, M, N, K, out=out, bias=b))
    outk = torch.empty(M, K, device=dev)
    t1 = timeit(lambda: tr.gemm(_C.GEMM_NN, g, w, M, K, N, out=outk))
    t2 = timeit(lambda: tr.gemm(_C.GEMM_TN, g, x, N, K, M))
    print(f"{name:5s} M={M} N={N} K={K}:  NT {t:7.1f} us {fl / t / 1e6:6.1f} TF | NN {t1:7.1f} us {fl / t1 / 1e6:6.1f} TF"
          f" | TN {t2:7.1f} us {fl / t2 / 1e6:6.1f} TF")

qkv = torch.randn(B, 64, 64, 3 * C, device=dev, requires_grad=True)
tab = torch.randn(225, 6, device=dev, requires_grad=True)
for shift in (0, 4):
    t = timeit(lambda: tr.window_attention(qkv.detach(), tab.detach(), 6, 8, shift, 30 ** -0.5))
    o = tr.window_attention(qkv, tab, 6, 8, shift, 30 ** -0.5)
    go = torch.randn_like(o)
    tb = timeit(lambda: torch.autograd.grad(o, (qkv, tab), go, retain_graph=True))
    fl = 4.0 * M * 64 * C
    print(f"wattn shift {shift}: fwd {t:7.1f} us ({fl / t / 1e6:5.1f} TF)  bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:5.1f} TF)")
x = torch.randn(M, C, device=dev, requires_grad=True)
gm, bt = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
t = timeit(lambda: tr.layer_norm(x.detach(), gm.detach(), bt.detach()))
y = tr.layer_norm(x, gm, bt)
gy = torch.randn_like(y)
tb = timeit(lambda: torch.autograd.grad(y, (x, gm, bt), gy, retain_graph=True))
by = M * C * 4
print(f"layernorm: fwd {t:6.1f} us ({2 * by / t / 1e3:6.0f} GB/s)  bwd {tb:6.1f} us ({3 * by / tb / 1e3:6.0f} GB/s)")
g = torch.randn(M, 540, device=dev)
t = timeit(lambda: tr.colsum(g))
print(f"colsum 540: {t:6.1f} us ({M * 540 * 4 / t / 1e3:6.0f} GB/s)")
