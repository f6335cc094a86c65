#!/usr/bin/env python
"""Dump the per-wave phase timeline of workgroup 0 of the conv kernel (needs a NEOSR_TIMELINE build:
`bash neosr_amd/csrc/build.sh -DNEOSR_TIMELINE` into a separate lib via NEOSR_AMD_OUT)."""
from __future__ import annotations

import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neosr_amd import _C  # noqa: E402
from neosr_amd.hip import ops  # noqa: E402


def run(name, H, W, K, N, CC, B=16, dgrad=False):
    lib = _C.load()
    x = torch.randn(B, H, W, CC, device="cuda")
    w = torch.randn(N, K, 3, 3, device="cuda") * 0.05
    out = torch.empty(B, H, W, N, device="cuda")
    tl = torch.zeros(4 * 64, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv3x3(x, w, None, out=out, k_in=K)
    torch.cuda.synchronize()
    lib.neosr_debug_set_timeline(tl.data_ptr())
    ops.conv3x3(x, w, None, out=out, k_in=K)
    torch.cuda.synchronize()
    lib.neosr_debug_set_timeline(None)
    t = tl.cpu().view(4, 64)
    nchunks = (K + 15) // 16
    print(f"== {name}: K={K} N={N} chunks={nchunks}  (cycles relative to wave start; s_memtime ticks)")
    for wv in range(4):
        r = t[wv]
        base = int(r[0])
        row = [f"gl0={int(r[1]) - base}"]
        for c in range(nchunks):
            b1, b2, gl, cp = (int(r[2 + 4 * c + j]) - base for j in range(4))
            row.append(f"[c{c} bar1@{b1} sst+bar2@{b2} gload@{gl} comp_end@{cp} (comp {cp - gl})]")
        row.append(f"epi@{int(r[62]) - base} end@{int(r[63]) - base}")
        print(f" wave{wv}: " + " ".join(row))


def run_glds(name, H, W, K, N, CC, B=16):
    """direct-to-LDS kernel: marks = start, first barrier, per chunk (issued, computed, barrier), epilogue"""
    lib = _C.load()
    x = torch.randn(B, H, W, CC, device="cuda")
    w = torch.randn(N, K, 3, 3, device="cuda") * 0.05
    pack = ops.conv3x3_pack_weights(w)
    out = torch.empty(B, H, W, N, device="cuda")
    tl = torch.zeros(4 * 64, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv3x3(x, w, None, out=out, k_in=K, w_pack=pack)
    torch.cuda.synchronize()
    lib.neosr_debug_set_timeline(tl.data_ptr())
    ops.conv3x3(x, w, None, out=out, k_in=K, w_pack=pack)
    torch.cuda.synchronize()
    lib.neosr_debug_set_timeline(None)
    t = tl.cpu().view(4, 64)
    nchunks = (K + 15) // 16
    print(f"== glds {name}: K={K} N={N} chunks={nchunks}")
    for wv in range(4):
        r = t[wv]
        base = int(r[0])
        row = [f"first_bar@{int(r[1]) - base}"]
        for c in range(nchunks - 1):
            i, cp, bar = (int(r[2 + 4 * c + j]) - base for j in range(3))
            row.append(f"[c{c} issued@{i} comp_end@{cp} (comp {cp - i}) bar@{bar} (wait {bar - cp})]")
        row.append(f"last_comp_end@{int(r[62]) - base} end@{int(r[63]) - base}")
        print(f" wave{wv}: " + " ".join(row))


def run_wino(name, H, W, K, N, CC, B=16):
    """Winograd kernel: marks = start, first barrier, per 32-channel chunk (half 0 done, half 1 done, barrier), exchange
    barrier, row pass done, end"""
    lib = _C.load()
    x = torch.randn(B, H, W, CC, device="cuda")
    w = torch.randn(N, K, 3, 3, device="cuda") * 0.05
    pack, wino = ops.conv3x3_pack_weights(w), ops.conv3x3_pack_wino(w)
    out = torch.empty(B, H, W, N, device="cuda")
    tl = torch.zeros(4 * 64, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv3x3(x, w, None, out=out, k_in=K, w_pack=pack, w_wino=wino)
    torch.cuda.synchronize()
    lib.neosr_debug_set_timeline(tl.data_ptr())
    ops.conv3x3(x, w, None, out=out, k_in=K, w_pack=pack, w_wino=wino)
    torch.cuda.synchronize()
    lib.neosr_debug_set_timeline(None)
    t = tl.cpu().view(4, 64)
    nchunks = (K + 31) // 32
    print(f"== winograd {name}: K={K} N={N} chunks={nchunks}")
    for wv in range(4):
        r = t[wv]
        base = int(r[0])
        row = [f"first_bar@{int(r[1]) - base}"]
        prev = int(r[1]) - base
        for c in range(nchunks):
            a, b_, bar = (int(r[2 + 3 * c + j]) - base for j in range(3))
            row.append(f"[c{c} h0 {a - prev} h1 {b_ - a} wait+bar {bar - b_}]")
            prev = bar
        row.append(f"exch_bar@{int(r[61]) - base} rowpass@{int(r[62]) - base} end@{int(r[63]) - base}")
        print(f" wave{wv}: " + " ".join(row))


def run_wino4(name, H, W, K, N, CC, B=16):
    """F(4x4,3x3) kernel (12 waves): marks = start, first barrier, per 32-channel chunk (hi MFMAs of the previous chunk done,
    transform done, lo MFMAs + refill issued, barrier passed), last hi MFMAs, exchange barrier, end"""
    lib = _C.load()
    x = torch.randn(B, H, W, CC, device="cuda")
    w = torch.randn(N, K, 3, 3, device="cuda") * 0.05
    pack, wino = ops.conv3x3_pack_weights(w), ops.conv3x3_pack_wino4(w)
    out = torch.empty(B, H, W, N, device="cuda")
    tl = torch.zeros(12 * 64, dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv3x3(x, w, None, out=out, k_in=K, w_pack=pack, w_wino4=wino)
    torch.cuda.synchronize()
    lib.neosr_debug_set_timeline(tl.data_ptr())
    ops.conv3x3(x, w, None, out=out, k_in=K, w_pack=pack, w_wino4=wino)
    torch.cuda.synchronize()
    lib.neosr_debug_set_timeline(None)
    t = tl.cpu().view(12, 64)
    nchunks = (K + 31) // 32
    t0 = int(t[:, 0].min())
    print(f"== winograd F(4x4) {name}: K={K} N={N} chunks={nchunks}  (cycles; per chunk: hiMFMA | transform | loMFMA | wait+barrier)")
    for wv in range(12):
        r = t[wv]
        base = int(r[0])
        row = [f"start+{base - t0} setup@{int(r[56]) - base} issued@{int(r[57]) - base} landed@{int(r[58]) - base} first_bar@{int(r[1]) - base}"]
        prev = int(r[1]) - base
        for c in range(nchunks):
            a, b_, m, bar = (int(r[2 + 4 * c + j]) - base for j in range(4))
            if c + 1 == nchunks:
                row.append(f"[c{c} {a - prev} | {b_ - a} | lo+bar {bar - b_}]")
            else:
                row.append(f"[c{c} {a - prev} | {b_ - a} | {m - b_} | {bar - m}]")
            prev = bar
        row.append(f"last_hi@{int(r[60]) - base} exch_written@{int(r[59]) - base} exch_bar@{int(r[61]) - base} colpass@{int(r[62]) - base} end@{int(r[63]) - base}")
        print(f" wave{wv:2d}: " + " ".join(row))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wino4":
        run_wino4("rdb.conv1", 64, 64, 64, 32, 192)
        run_wino4("rdb.conv4", 64, 64, 160, 32, 192)
        raise SystemExit
    if len(sys.argv) > 1 and sys.argv[1] == "wino":
        run_wino("rdb.conv1", 64, 64, 64, 32, 192)
        run_wino("rdb.conv4", 64, 64, 160, 32, 192)
        run_wino("rdb.conv5", 64, 64, 192, 64, 192)
        raise SystemExit
    run_glds("rdb.conv1", 64, 64, 64, 32, 192)
    run_glds("rdb.conv4", 64, 64, 160, 32, 192)
    run_glds("rdb.conv5", 64, 64, 192, 64, 192)
    run("rdb.conv1", 64, 64, 64, 32, 192)
    run("rdb.conv5", 64, 64, 192, 64, 192)
    run("conv_hr@256", 256, 256, 64, 64, 64)
