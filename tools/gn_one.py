"""GroupNorm(+SiLU) forward / backward timing at network sizes through the C-ABI; prints effective HBM GB/s (algorithmic bytes).
usage: python tools/gn_one.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
P = _lib.ptr; S = _lib.stream_ptr
def run(B, H, W, C, mode=0, reps=10):
    G = min(C // 4, 32)
    x = torch.randn(B, H, W, C, device="cuda"); gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    Ho, Wo = (H // 2, W // 2) if mode == 1 else ((H * 2, W * 2) if mode == 2 else (H, W))
    y = torch.empty(B, Ho, Wo, C, device="cuda"); dy = torch.randn(B, Ho, Wo, C, device="cuda"); dx = torch.empty_like(x)
    stats = torch.empty(B, G, 2, device="cuda"); red = torch.empty(B, G, 2, device="cuda"); scratch = torch.empty(B * 256 * C * 4, device="cuda")
    f = lambda: _lib.check(lib.buddy_groupnorm_act(P(x), P(gamma), P(beta), P(y), P(stats), P(scratch), B, H, W, C, G, mode, 1, S()))
    b = lambda: _lib.check(lib.buddy_groupnorm_act_bwd(P(x), P(gamma), P(beta), P(stats), P(dy), P(dx), P(scratch), P(red), B, H, W, C, G, mode, 1, S()))
    for fn, name, passes in ((f, "fwd (stats + apply)", 2 * x.numel() + y.numel()), (b, "bwd (sums + apply)", 2 * x.numel() + 2 * dy.numel() + x.numel())):
        fn(); fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
        print(f"GN B{B} {H}x{W} C{C} mode{mode} {name}: {dt*1e3:.3f} ms  {passes*4/dt/1e9:.0f} GB/s")
run(8, 512, 256, 128); run(8, 512, 256, 256); run(8, 512, 256, 384); run(8, 256, 128, 256); run(8, 512, 256, 128, 1); run(8, 256, 128, 256, 2)
