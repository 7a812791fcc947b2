"""ONE shape of the Winograd-domain batched GEMM (csrc/wgemm.hip), a few launches (for rocprofv3 --pmc passes and kernel experiments).
usage: python tools/wgemm_one.py Mt N K [positions] [reps] [bf16x3|f16x2]   (f16x2: Mt = 8 utterances x Mt / 8 tiles)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
P = _lib.ptr; S = _lib.stream_ptr
Mt, N, K = (int(v) for v in sys.argv[1:4]); nb = int(sys.argv[4]) if len(sys.argv) > 4 else 64; reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
A = torch.randn(nb, Mt, K, device="cuda"); Bt = torch.randn(nb, N, K, device="cuda"); Cm = torch.empty(nb, Mt, N, device="cuda")
U3 = torch.empty(nb * N * K * 6 // 4, dtype=torch.int32, device="cuda")
_lib.check(lib.buddy_wgemm_pack_weights(P(Bt), U3.data_ptr(), nb, N, K, S()))
mode = sys.argv[6] if len(sys.argv) > 6 else "bf16x3"
f = lambda: _lib.check(lib.buddy_gemm_winograd_domain_bf16x3(P(A), U3.data_ptr(), P(Cm), Mt, N, K, nb, S()))
if mode == "f16x2":
    U2 = torch.empty(int(lib.buddy_wgemm_f16x2_packed_bytes(nb, N, K)) // 4, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_wgemm_f16x2_pack_weights(P(Bt), U2.data_ptr(), nb, N, K, S()))
    vmax = torch.empty(8, 64, 32, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_abs_max_bits(P(A), nb, 8, (Mt // 8) * K, vmax.data_ptr(), S()))
    f = lambda: _lib.check(lib.buddy_gemm_winograd_domain_f16x2(P(A), U2.data_ptr(), P(Cm), Mt, N, K, nb, vmax.data_ptr(), Mt // 8, S()))
f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(reps): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
print(f"wgemm {mode} P={nb} Mt={Mt} N={N} K={K}: {dt*1e3:.3f} ms {2.0*nb*Mt*N*K/dt/1e12:.1f} TF-eq {(6.0 if mode == 'f16x2' else 12.0)*nb*Mt*N*K/dt/1e12:.0f} TF executed "
      f"{(Mt*K+Mt*N)*nb*4/dt/1e9:.0f} GB/s", flush=True)
