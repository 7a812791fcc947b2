"""Does replaying the network's forward + VJP as ONE captured graph beat the eager launches?  (The launches are ~520 per step; a kernel trace shows
4-19 us idle between consecutive eager kernels.)  Times fwd + vjp eager and as a torch.cuda.CUDAGraph replay of the same C-ABI calls on fixed buffers.
usage: python tools/graph_probe.py [B] [L]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
from tests.test_hip_network import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
lib = _lib.require_gpu(); P = _lib.ptr
net = build(128, 510, 128, 0)
net.reserve(B, L, True) if hasattr(net, "reserve") else None
h = net._get_handle()
x = (0.1 * torch.randn(B, L)).cuda(); cn = torch.full((B,), -0.7).cuda()
cin = torch.ones(B).cuda(); cskip = torch.zeros(B).cuda(); cout = torch.ones(B).cuda()
y = torch.empty(B, L).cuda(); cot = torch.randn(B, L).cuda(); gx = torch.empty(B, L).cuda()
def step(s):
    _lib.check(lib.buddy_ncsnpp_forward(h, P(x), P(cn), P(cin), P(cskip), P(cout), P(y), B, L, 1, s))
    _lib.check(lib.buddy_ncsnpp_vjp(h, P(cot), P(gx), s))
def timed(f, n=5):
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
te = timed(lambda: step(torch.cuda.current_stream().cuda_stream))
y0, g0 = y.clone(), gx.clone()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step(side.cuda_stream); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        step(side.cuda_stream)
tg = timed(lambda: g.replay())
torch.cuda.synchronize()
print(f"fwd + vjp B={B} L={L}: eager {te:.2f} ms, graph replay {tg:.2f} ms; results equal: {torch.equal(y, y0) and torch.equal(gx, g0)}")
