#!/usr/bin/env python
"""The F(4x4-tile) weight gradient of conv3x3_wgrad_wino4_kernel in float64 (numpy, CPU): the 3x3 tap gradient of one
(cout, cin) pair from 4x4 gradient tiles and 6x6 input patches,
    dW = C [ sum_tiles (G'' dy G''^T) .* (B^T d B) ] C^T,
B^T = the 6x6 input transform of F(4x4,3x3), G'' = the F(3,4) filter transform with its row scales (1/4, -1/6, -1/6, 1/24,
1/24, 1) taken out (small integers), C = A'^T diag(scales) — against the direct sum dW[a, b] = sum_p dy[p] x[p + (a, b)].
(Re-created in round 4: the round-3 copy lived in the untracked experiments/ directory.)"""
import numpy as np

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
               [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
pts = [0.0, 1.0, -1.0, 2.0, -2.0]
scales = np.array([1 / 4, -1 / 6, -1 / 6, 1 / 24, 1 / 24, 1.0])
G2 = np.array([[p ** k for k in range(4)] for p in pts] + [[0, 0, 0, 1]], np.float64)        # G'' (6 x 4): integers
AT = np.array([[p ** i for p in pts] + [0.0] for i in range(2)] + [[p ** 2 for p in pts] + [1.0]], np.float64)   # A'^T (3 x 6)
Cm = AT * scales[None, :]
assert np.allclose(G2, np.round(G2)), "G'' is integral"

rng = np.random.default_rng(1)
H = W = 16
x = rng.standard_normal((H + 2, W + 2))
dy = rng.standard_normal((H, W))
ref = np.array([[(dy * x[a:a + H, b:b + W]).sum() for b in range(3)] for a in range(3)])
acc = np.zeros((6, 6))
for ty in range(0, H, 4):
    for tx in range(0, W, 4):
        acc += (G2 @ dy[ty:ty + 4, tx:tx + 4] @ G2.T) * (BT @ x[ty:ty + 6, tx:tx + 6] @ BT.T)
got = Cm @ acc @ Cm.T
print("max |dW - direct| =", np.abs(got - ref).max(), " (|dW| max", np.abs(ref).max(), ")")
assert np.abs(got - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
print("ok")
