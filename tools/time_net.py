import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests.test_hip_network import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
net = build(128, 510, 128, 0)
x = (0.1 * torch.randn(B, L)).cuda()
cn = torch.full((B,), -0.7).cuda()
print("arena GB fwd-only", net.arena_bytes(B, L, False) / 1e9, "with vjp", net.arena_bytes(B, L, True) / 1e9)
with torch.no_grad():
    for _ in range(2): y = net(x, cn)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): y = net(x, cn)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
print(f"fwd B={B} L={L}: {dt*1e3:.2f} ms  -> {1.2501e12*B/dt/1e12:.1f} TFLOP/s", float(y.abs().max()))
xg = x.clone().requires_grad_(True)
cot = torch.randn(B, L).cuda()
for _ in range(2):
    y = net(xg, cn); g, = torch.autograd.grad(y, xg, cot)
torch.cuda.synchronize(); t = time.time()
for _ in range(5):
    y = net(xg, cn); g, = torch.autograd.grad(y, xg, cot)
torch.cuda.synchronize(); dt = (time.time() - t) / 5
print(f"fwd+vjp B={B} L={L}: {dt*1e3:.2f} ms -> {2*1.2501e12*B/dt/1e12:.1f} TFLOP/s", float(g.abs().max()))
