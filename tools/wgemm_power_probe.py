"""Is the batched Winograd-domain GEMM power-limited?  The same launch on random, sign-constant and zero operands (identical instruction stream and memory
traffic; only the switching activity differs), in both split arithmetics.  usage: python tools/wgemm_power_probe.py [Mt N K]   (Mt = 8 x tiles per utterance)"""
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
    U2 = torch.empty(int(lib.buddy_wgemm_f16x2_packed_bytes(nb, N, K)) // 4, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_wgemm_f16x2_pack_weights(P(Bt), U2.data_ptr(), nb, N, K, S()))
    vmax = torch.empty(8, 64, 32, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_abs_max_bits(P(A), nb, 8, (Mt // 8) * K, vmax.data_ptr(), S()))
    f2 = lambda: _lib.check(lib.buddy_gemm_winograd_domain_f16x2(P(A), U2.data_ptr(), P(Cm), Mt, N, K, nb, vmax.data_ptr(), Mt // 8, S()))
    for _ in range(3): f2()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f2()
    torch.cuda.synchronize(); d2 = (time.perf_counter() - t) / 20
    print(f"{name:22s} Mt={Mt} N={N} K={K}: bf16x3 {dt*1e3:.3f} ms {12.0*nb*Mt*N*K/dt/1e12:.0f} TF | f16x2 {d2*1e3:.3f} ms {6.0*nb*Mt*N*K/d2/1e12:.0f} TF "
          f"{4.0*nb*Mt*(N+K)/d2/1e9:.0f} GB/s", flush=True)
