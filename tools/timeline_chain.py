#!/usr/bin/env python
"""Per-wave timeline of workgroup 0 of conv3x3_wino4_chain_kernel (NEOSR_TIMELINE build in experiments/tl:
NEOSR_AMD_OUT=$PWD/experiments/tl bash neosr_amd/csrc/build.sh -DNEOSR_TIMELINE; run with NEOSR_AMD_LIB=experiments/tl/libneosr_amd.so).
Marks per layer: 0 start, 1 drained (vmcnt 0), 2 first barrier passed, 3 loop start, 4 loop end, 5 exchange written,
6 exchange barrier passed, 7 stores issued."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neosr_amd import _C
from neosr_amd.archs import build_network

lib = _C.load()
torch.manual_seed(0)
net = build_network({"type": "esrgan", "scale": 4, "num_block": 2}).cuda().train()
x = torch.rand(16, 3, 64, 64, device="cuda")
for _ in range(3):
    with torch.no_grad():
        net(x)
torch.cuda.synchronize()
tl = torch.zeros(12 * 16 * 16, dtype=torch.int64, device="cuda")
lib.neosr_debug_set_timeline(tl.data_ptr())
with torch.no_grad():
    net(x)
torch.cuda.synchronize()
lib.neosr_debug_set_timeline(None)
t = tl.cpu().view(12, 16, 16)
base = int(t[:, 0, 0].min())
names = ["start", "drained", "bar1", "loop", "loopend", "exch", "exbar", "stored", "entry", "rec"]
for l in range(15):
    print(f"-- layer {l} (conv{l % 5 + 1})")
    for w in (0, 1, 4, 5, 8, 11):
        r = t[w, l][:10]
        vals = [int(v) - base if int(v) else None for v in r]
        print(f"   wave{w:2d}: " + " ".join(f"{n}@{v}" for n, v in zip(names, vals)))
print("== chunk starts of wave 0 / wave 8 per layer (ticks since the layer's loop mark)")
for l in range(15):
    for w in (0, 8):
        r = t[w, l]
        print(f" layer {l:2d} wave {w}: loop@0 " + " ".join(f"c{c}@{int(r[10 + c]) - int(r[3])}" for c in range(6) if int(r[10 + c])) + f" loopend@{int(r[4]) - int(r[3])}")
# summary: per layer, max over waves of each mark, as deltas
print("== per layer (max over waves), cycles: start->bar1, bar1->loop, loop, loopend->exbar, exbar->stored(max fin), total")
prev_end = None
for l in range(15):
    m = t[:, l, :].clone()
    st = int(m[:, 0].min()) - base
    bar1 = int(m[:, 2].max()) - base
    loop = int(m[:, 3].max()) - base
    le = int(m[:, 4].max()) - base
    exb = int(m[:, 6].max()) - base
    sto = int(m[4:, 7].max()) - base
    nxt = int(t[:, l + 1, 0].min()) - base if l + 1 < 15 else sto
    print(f" layer {l:2d}: start@{st} | {bar1 - st} | {loop - bar1} | {le - loop} | {exb - le} | {sto - exb} | layer span {nxt - st}")
