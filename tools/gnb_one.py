"""GroupNorm(+SiLU) backward (sums + apply) at ONE shape, for rocprofv3 (tools/prof_py.sh tools/gnb_one.py 8 B H W C).  Prints algorithmic GB/s of
the pair; the per-kernel times come from the profiler.  usage: python tools/gnb_one.py [B H W C]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu(); P = _lib.ptr; S = _lib.stream_ptr
B, H, W, C = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 512, 256, 128)
G = min(C // 4, 32)
x = torch.randn(B, H, W, C, device="cuda"); gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
stats = torch.empty(B, G, 2, device="cuda"); red = torch.empty(B, G, 2, device="cuda"); scratch = torch.empty(B * 256 * C * 4, device="cuda")
_lib.check(lib.buddy_groupnorm_act(P(x), P(gamma), P(beta), P(y), P(stats), P(scratch), B, H, W, C, G, 0, 1, S()))
b = lambda: _lib.check(lib.buddy_groupnorm_act_bwd(P(x), P(gamma), P(beta), P(stats), P(dy), P(dx), P(scratch), P(red), B, H, W, C, G, 0, 1, S()))
b(); b(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): b()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"GN bwd B{B} {H}x{W} C{C}: {dt*1e3:.3f} ms; tensor {x.numel()*4/1e6:.0f} MB; sums+apply {5*x.numel()*4/dt/1e9:.0f} GB/s")
a = torch.empty_like(x)
torch.add(x, dy, out=a); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): torch.add(x, dy, out=a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print(f"torch add (2 reads + 1 write) on the same tensors: {dt*1e3:.3f} ms  {3*x.numel()*4/dt/1e9:.0f} GB/s")
