import sys, time, torch
sys.path.insert(0, ".")
from neosr_amd import _C
from neosr_amd.archs import build_network
import torch.nn.functional as F
torch.manual_seed(0)
net = build_network({"type": "esrgan", "scale": 4}).to("cuda").train()
x = torch.rand(16, 3, 64, 64, device="cuda"); gt = torch.rand(16, 3, 256, 256, device="cuda")
for fast in (False, True, False, True):
    _C.set_fast_matmul(fast)
    for _ in range(3):
        net.zero_grad(set_to_none=True); F.l1_loss(net(x), gt).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        net.zero_grad(set_to_none=True); F.l1_loss(net(x), gt).backward()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"fast={fast}: fwd+bwd {1e3*dt:.2f} ms  ({16/dt:.1f} patches/s without optimizer)")
