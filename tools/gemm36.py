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
    U3 = torch.empty(nb * N * K * 6 // 4, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_wgemm_pack_weights(P(A) and P(Bt), U3.data_ptr(), nb, N, K, S()))
    out = []
    for f in (lambda: _lib.check(lib.buddy_gemm_winograd_domain(P(A), P(Bt), P(Cm), Mt, N, K, nb, S())),
              lambda: _lib.check(lib.buddy_gemm_winograd_domain_bf16x3(P(A), U3.data_ptr(), P(Cm), Mt, N, K, nb, S()))):
        f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / reps)
    d1, d3 = out
    print(f"batch={nb} Mt={Mt:6d} N={N:4d} K={K:4d}: fp32 MFMA {d1*1e3:7.3f} ms {2.0*nb*Mt*N*K/d1/1e12:6.1f} TF | bf16x3 {d3*1e3:7.3f} ms "
          f"{2.0*nb*Mt*N*K/d3/1e12:6.1f} TF-equivalent ({(Mt*K+Mt*N)*nb*4/d3/1e9:5.0f} GB/s)  x{d1/d3:.2f}", flush=True)
for s in ((65536, 256, 256), (65536, 128, 128), (65536, 128, 384), (65536, 384, 128), (65536, 256, 128), (65536, 128, 256),
          (16384, 256, 256), (16384, 256, 512), (16384, 512, 256), (4096, 256, 256), (1024, 256, 256)):
    run(*s)
for s in ((29584, 256, 256), (29584, 128, 128), (29584, 128, 384), (29584, 384, 128), (29584, 256, 128), (29584, 128, 256),
          (7568, 256, 256), (7568, 256, 512), (7568, 512, 256), (1936, 256, 256), (1936, 256, 512)):
    run(*s, nb=64)
