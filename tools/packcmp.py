"""debug: neosr_conv3x3_pack_many (kind 2 = Winograd F(4x4) images) of two library builds on a mixed set of HAT shapes"""
import sys, os, subprocess
import numpy as np
code = r'''
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from neosr_amd import _C
lib = _C.load()
torch.manual_seed(0)
shapes = [(60, 180), (180, 60), (180, 180), (64, 180), (180, 64), (3, 64)] * 7
items, keep, res = [], [], {}
for i, (co, ci) in enumerate(shapes):
    for mode in (0, 1):
        w = torch.randn(co, ci, 3, 3, device="cuda")
        N, K = (co, ci) if mode == 0 else (ci, co)
        nb = lib.neosr_conv3x3_pack_wino4_bytes(N, K)
        dst = torch.full((nb // 4,), 7.0, device="cuda")
        keep.append((w, dst))
        items.append(_C.PackItem(w=w.data_ptr(), dst=dst.data_ptr(), w_cout=co, w_cin=ci, mode=mode, kind=2))
arr = (_C.PackItem * len(items))(*items)
rc = lib.neosr_conv3x3_pack_many(arr, len(items), None)
torch.cuda.synchronize()
for i, (w, dst) in enumerate(keep):
    res["u%03d" % i] = dst.cpu().numpy()
np.savez(sys.argv[1], **res)
'''
for tag, path in (("prev", "experiments/prev/libneosr_amd.so"), ("cur", "neosr_amd/lib/libneosr_amd.so")):
    subprocess.run([sys.executable, "-c", code, "/tmp/pack_%s.npz" % tag], env=dict(os.environ, NEOSR_AMD_LIB=os.path.join(os.getcwd(), path)), check=True)
a, b = np.load("/tmp/pack_prev.npz"), np.load("/tmp/pack_cur.npz")
worst = 0
for k in a.files:
    d = np.abs(a[k] - b[k]).max() / np.abs(a[k]).max()
    if d > 1e-6: print(k, "REL DIFF", d, "n", a[k].size, "bad", int((np.abs(a[k]-b[k]) > 1e-6).sum()))
    worst = max(worst, d)
print("worst rel diff over", len(a.files), "images:", worst)
