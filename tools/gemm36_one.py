"""ONE shape of the 36-batch Winograd-domain GEMM, a few launches (for rocprofv3 --pmc passes).  usage: python tools/gemm36_one.py Mt N K [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
P = _lib.ptr; S = _lib.stream_ptr
Mt, N, K = (int(v) for v in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
A = torch.randn(36, Mt, K, device="cuda"); Bt = torch.randn(36, N, K, device="cuda"); Cm = torch.empty(36, Mt, N, device="cuda")
f = lambda: _lib.check(lib.buddy_gemm_winograd_domain(P(A), P(Bt), P(Cm), Mt, N, K, 36, S()))
f(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
print(f"Mt={Mt} N={N} K={K}: {dt*1e3:.3f} ms {2.0*36*Mt*N*K/dt/1e12:.1f} TF")
