#!/usr/bin/env python
"""Clock marks of ONE 64 x 64 tile of the fused HAT attention backward (workgroup 0, thread 0; a -DFATTN_TL build of the
library: `cd neosr_amd/csrc && NEOSR_AMD_OUT=../../experiments/tl bash build.sh -DFATTN_TL`, then
NEOSR_AMD_LIB=experiments/tl/libneosr_amd.so python tools/timeline_fattn.py [B]).  GPU box only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neosr_amd.hip import transformer as tr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C, heads = 180, 6
qkv = torch.randn(B, 64, 64, 3 * C, device="cuda", requires_grad=True)
tab = torch.randn(31 * 31, heads, device="cuda", requires_grad=True)
o = tr.flash_window_attention(qkv, tab, heads, 16, 0, 30 ** -0.5)
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
assert lib.neosr_debug_fattn_timeline(buf) == 0
t = list(buf)
names = {1: "tile top", 2: "barrier 1 passed", 3: "K / V rows stored + barrier 2", 4: "next rows requested", 30: "S, dP products done",
         31: "P, dS written", 5: "barrier 3 passed", 6: "bias bins walked", 7: "dV / dK product done", 8: "dQ product done", 9: "next tile top"}
order = [1, 2, 3, 4, 30, 31, 5, 6, 7, 8, 9]
print(f"whole kernel (workgroup 0): {t[20] - t[0]} ticks; query blocks end at {[t[10 + q] - t[0] for q in range(4)]}")
prev = t[1]
for k in order[1:]:
    print(f"  {names[k]:32s} +{t[k] - prev:7d}   (at {t[k] - t[1]})")
    prev = t[k]
