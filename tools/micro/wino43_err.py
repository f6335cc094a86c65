#!/usr/bin/env python
"""Rounding of the Winograd forms against a float64 convolution (numpy, CPU): direct fp32, F(2x2,3x3) and F(4x4,3x3) with
the weights transformed in float64 and rounded once (what conv_pack_wino*_kernel do) and everything else in fp32 — the
figures DESIGN.md §3 quotes (direct ~1e-6, F(2x2) ~4e-7, F(4x4) ~5e-6 of the output scale).  (Re-created in round 4:
the round-3 copy lived in the untracked experiments/ directory.)"""
import numpy as np

rng = np.random.default_rng(0)
K, N, H, W = 64, 32, 32, 32
x = rng.standard_normal((K, H + 2, W + 2)).astype(np.float32)
w = (rng.standard_normal((N, K, 3, 3)) / np.sqrt(9 * K)).astype(np.float32)


def direct(x, w, dt):
    y = np.zeros((N, H, W), dt)
    for a in range(3):
        for b in range(3):
            y += np.einsum("nk,khw->nhw", w[:, :, a, b].astype(dt), x[:, a:a + H, b:b + W].astype(dt))
    return y


def wino(x, w, m):
    if m == 2:
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
    else:
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                      [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
    a = m + 2
    U = np.einsum("ia,nkab,jb->nkij", G, w.astype(np.float64), G).astype(np.float32)   # float64, rounded once
    BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)
    y = np.zeros((N, H, W), np.float32)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = x[:, ty:ty + a, tx:tx + a]
            V = np.einsum("ia,kab,jb->kij", BT32, d, BT32).astype(np.float32)
            Mm = np.einsum("nkij,kij->nij", U, V).astype(np.float32)
            y[:, ty:ty + m, tx:tx + m] = np.einsum("ia,nab,jb->nij", AT32, Mm, AT32)
    return y


ref = direct(x, w, np.float64)
scale = np.abs(ref).max()
for name, y in (("direct fp32", direct(x, w, np.float32)), ("F(2x2,3x3)", wino(x, w, 2)), ("F(4x4,3x3)", wino(x, w, 4))):
    print(f"{name:12s} max |err| / max |y| = {np.abs(y - ref).max() / scale:.2e}   rel L2 = "
          f"{np.linalg.norm(y - ref) / np.linalg.norm(ref):.2e}")
