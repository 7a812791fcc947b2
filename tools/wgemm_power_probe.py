"""Is the batched bf16x3 GEMM power-limited?  The same launch on random, sign-constant and zero operands (identical instruction stream and memory traffic;
only the switching activity differs).  usage: python tools/wgemm_power_probe.py [Mt N K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu(); P = _lib.ptr; S = _lib.stream_ptr
Mt, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (7568, 256, 256)
nb = 64
Cm = torch.empty(nb, Mt, N, device="cuda"); U3 = torch.empty(nb * N * K * 6 // 4, dtype=torch.int32, device="cuda")
for name, gen in (("random N(0,1)", lambda *s: torch.randn(*s, device="cuda")), ("abs(random)", lambda *s: torch.randn(*s, device="cuda").abs()),
                  ("zeros", lambda *s: torch.zeros(*s, device="cuda")), ("random N(0,1) again", lambda *s: torch.randn(*s, device="cuda"))):
    A = gen(nb, Mt, K); Bt = gen(nb, N, K)
    _lib.check(lib.buddy_wgemm_pack_weights(P(Bt), U3.data_ptr(), nb, N, K, S()))
    f = lambda: _lib.check(lib.buddy_gemm_winograd_domain_bf16x3(P(A), U3.data_ptr(), P(Cm), Mt, N, K, nb, S()))
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"{name:22s} Mt={Mt} N={N} K={K}: {dt*1e3:.3f} ms  {12.0*nb*Mt*N*K/dt/1e12:.0f} TF bf16", flush=True)
