import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from neosr_amd.archs import build_network
torch.manual_seed(0)
net = build_network({"type": sys.argv[1] if len(sys.argv) > 1 else "hat_l", "drop_path_rate": 0.0}).cuda().train()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
x = torch.rand(B, 3, 64, 64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def whole():
    y = net(x); y.sum().backward()
def halves():
    main = torch.cuda.current_stream()
    outs = []
    for s, xs in ((s1, x[: B // 2]), (s2, x[B // 2:])):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs.append(net(xs))
    for s in (s1, s2): main.wait_stream(s)
    y = torch.cat(outs); y.sum().backward()
    for s in (s1, s2): main.wait_stream(s)
def fwd_whole():
    with torch.no_grad(): net(x)
def fwd_halves():
    main = torch.cuda.current_stream()
    with torch.no_grad():
        for s, xs in ((s1, x[: B // 2]), (s2, x[B // 2:])):
            s.wait_stream(main)
            with torch.cuda.stream(s): net(xs)
    for s in (s1, s2): main.wait_stream(s)
def t(fn, n=5):
    for _ in range(2): fn(); net.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn(); net.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, fn in (("fwd whole", fwd_whole), ("fwd halves", fwd_halves), ("fwd+bwd whole", whole), ("fwd+bwd halves", halves)):
    print(f"{name:16s} {t(fn):8.2f} ms", flush=True)
