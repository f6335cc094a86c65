#!/usr/bin/env python
"""HAT 16x16-window attention (self and overlapping) forward / backward timing at the hat_l shape (GPU box only)."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from neosr_amd.hip import transformer as tr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C, heads = 180, 6
dev = "cuda"


def timeit(fn, n=50):
    for _ in range(30):   # (the first calls of a process run ~2x slower: clocks and caches)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


qkv = torch.randn(B, 64, 64, 3 * C, device=dev, requires_grad=True)
for ks, shift in ((16, 0), (16, 8), (24, 0)):
    L = 16 + ks - 1
    tab = torch.randn(L * L, heads, device=dev, requires_grad=True)
    t = timeit(lambda: tr.flash_window_attention(qkv.detach(), tab.detach(), heads, ks, shift, 30 ** -0.5))
    o = tr.flash_window_attention(qkv, tab, heads, ks, shift, 30 ** -0.5)
    go = torch.randn_like(o)
    tb = timeit(lambda: torch.autograd.grad(o, (qkv,), go, retain_graph=True))
    print(f"fattn ks {ks} shift {shift}: fwd {t:7.1f} us  bwd {tb:7.1f} us")
