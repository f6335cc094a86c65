#!/usr/bin/env python
"""Clock marks of the SwinIR 8x8 window-attention backward (workgroup 0, waves 0 and 2; a -DWATTN_TL build of the library:
`cd neosr_amd/csrc && NEOSR_AMD_OUT=../../experiments/tl bash build.sh -DWATTN_TL`, run under
NEOSR_AMD_LIB=experiments/tl/libneosr_amd.so).  GPU box only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neosr_amd.hip import transformer as tr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C, heads = 180, 6
qkv = torch.randn(B, 64, 64, 3 * C, device="cuda", requires_grad=True)
tab = torch.randn(225, heads, device="cuda", requires_grad=True)
o = tr.window_attention(qkv, tab, heads, 8, 0, 30 ** -0.5)
go = torch.randn_like(o)
for _ in range(20):
    torch.autograd.grad(o, (qkv,), go, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    torch.autograd.grad(o, (qkv,), go, retain_graph=True)
e1.record()
torch.cuda.synchronize()
print(f"B = {B}: backward call {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
lib = ctypes.CDLL(os.environ["NEOSR_AMD_LIB"])
buf = (ctypes.c_ulonglong * 64)()
assert lib.neosr_debug_wattn_timeline(buf) == 0
names = [None, "rows requested, tables built (their loads waited for)", "q stored", "k, v, dO stored + delta", "barrier", "S product",
         "P written", "dP product", "barrier (P complete)", "dV product + stores (waves 2, 3)", "barrier", "dS in place", "barrier",
         "bias bins", "dQ / dK product + stores"]
for w in (0, 1):
    t = list(buf)[32 * w: 32 * w + 16]
    print(f" wave {2 * w}: total {t[14] - t[0]} ticks")
    for i in range(1, 15):
        print(f"   {names[i]:52s} +{t[i] - t[i - 1]:6d}  (at {t[i] - t[0]})")
