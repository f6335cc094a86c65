#!/usr/bin/env python
"""Scan the ISA of every kernel of the library for the three patterns that cost the attention kernels 10-25 % in round 6
(DESIGN §0.1): (a) `s_waitcnt lgkmcnt(0)` between two LDS reads (a chain of LDS round trips where a batch was meant),
(b) `s_waitcnt vmcnt(0)` a few instructions in front of a global / buffer load (a load that waits for the previous one),
(c) a global load right behind `s_cbranch_execz` (`cond ? *p : 0` compiled as a branch around the load).  No GPU needed:

  mkdir -p /tmp/asm && for f in neosr_amd/csrc/*.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm \\
      --cuda-device-only -S $f -o /tmp/asm/$(basename $f .hip).s; done
  python tools/isa_scan.py /tmp/asm

A count is a lead, not a verdict: hand-scheduled kernels (the chain kernel, the Winograd weight gradient) wait on purpose;
read the kernel with tools/isa_summary.py before changing it."""
import glob, re, sys

res = []
for f in sorted(glob.glob(sys.argv[1].rstrip("/") + "/*.s")):
    name, lines = None, []
    for raw in open(f):
        m = re.match(r"^(_Z\w+):\s", raw)
        if m:
            name, lines = m.group(1), []
            continue
        if name is None:
            continue
        x = raw.strip()
        lines.append(x)
        if not x.startswith("s_endpgm"):
            continue
        ser = vm0 = brload = 0
        for i, x in enumerate(lines):
            if x.startswith("s_waitcnt") and "lgkmcnt(0)" in x:
                if any(y.startswith("ds_read") for y in lines[max(0, i - 3):i]) and any(y.startswith("ds_read") for y in lines[i + 1:i + 8]):
                    ser += 1
            if x.startswith("s_waitcnt") and "vmcnt(0)" in x:
                if any(y.startswith(("global_load", "buffer_load")) for y in lines[i + 1:i + 12]):
                    vm0 += 1
            if x.startswith(("global_load", "buffer_load")) and " lds" not in x:
                if any(y.startswith("s_cbranch_execz") for y in lines[max(0, i - 4):i]):
                    brload += 1
        res.append((f.split("/")[-1], name, len(lines), ser, vm0, brload))
        name = None
print(f"{'file':22s} {'lines':>6s} {'LDS chain':>9s} {'load waits load':>15s} {'branchy load':>12s}  kernel")
for r in sorted(res, key=lambda r: -(r[3] + 4 * r[4] + r[5])):
    if r[3] + r[4] + r[5] > 6:
        print(f"{r[0]:22s} {r[2]:6d} {r[3]:9d} {r[4]:15d} {r[5]:12d}  {r[1][:100]}")
