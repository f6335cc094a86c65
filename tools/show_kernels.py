#!/usr/bin/env python
"""stdin: one bench.py JSON line -> value, ms/step and the per-class kernel table of its roofline record"""
import json
import sys

d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", d["value"], "patches/s", d["ms_per_step"], "ms/step")
for k, v in d["roofline"]["kernels"].items():
    print("   %-64s launches %5d  avg %8.2f us  total %8.3f ms  kernel launches %s" % (
        k[:64], v["launches"], v["avg_us"], v["total_ms"], v.get("kernel_launches")))
