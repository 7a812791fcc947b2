"""The batched GEMM pass of the three-pass Winograd convolutions in isolation, per layer shape (B=8): 36 positions x 65536 / 16384 / ... tiles
(F(4x4,3x3)) and 64 positions x 29584 / 7568 / 1936 tiles (F(6x6,3x3), levels 0-2).  usage: python tools/gemm36.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
P = _lib.ptr; S = _lib.stream_ptr
def run(Mt, N, K, reps=5, nb=36):
    A = torch.randn(nb, Mt, K, device="cuda"); Bt = torch.randn(nb, N, K, device="cuda"); Cm = torch.empty(nb, Mt, N, device="cuda")
    f = lambda: _lib.check(lib.buddy_gemm_winograd_domain(P(A), P(Bt), P(Cm), Mt, N, K, nb, S()))
    f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    print(f"batch={nb} Mt={Mt:6d} N={N:4d} K={K:4d}: {dt*1e3:7.3f} ms  {2.0*nb*Mt*N*K/dt/1e12:6.1f} TF  ({(Mt*K+Mt*N)*nb*4/dt/1e9:5.0f} GB/s)")
for s in ((65536, 256, 256), (65536, 128, 128), (65536, 128, 384), (65536, 384, 128), (65536, 256, 128), (65536, 128, 256),
          (16384, 256, 256), (16384, 256, 512), (16384, 512, 256), (4096, 256, 256), (1024, 256, 256)):
    run(*s)
for s in ((29584, 256, 256), (29584, 128, 128), (29584, 128, 384), (29584, 384, 128), (29584, 256, 128), (29584, 128, 256),
          (7568, 256, 256), (7568, 256, 512), (7568, 512, 256), (1936, 256, 256), (1936, 256, 512)):
    run(*s, nb=64)
