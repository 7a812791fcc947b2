"""Can an HBM-bound kernel of one stream run BESIDE the MFMA-bound Winograd-domain GEMM of another stream?  Times the 36-batch GEMM alone, a
chain of streaming kernels alone, and both together on two streams.  Run with BUDDY_GEMM_LDS_PAD=0 / 24576 / 49152 (caps the GEMM's resident
workgroups per CU).  usage: python tools/overlap_probe.py [Mt N K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
P = _lib.ptr
Mt, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (65536, 256, 256)
A = torch.randn(36, Mt, K, device="cuda"); Bt = torch.randn(36, N, K, device="cuda"); Cm = torch.empty(36, Mt, N, device="cuda")
U3 = torch.empty(36 * N * K * 6 // 4, dtype=torch.int32, device="cuda")
_lib.check(lib.buddy_wgemm_pack_weights(P(Bt), U3.data_ptr(), 36, N, K, _lib.stream_ptr()))
BF = os.environ.get("BUDDY_GEMM", "bf16x3") != "fp32"
x = torch.randn(128 * 2 ** 20 // 4 * 4, device="cuda"); y = torch.empty_like(x)       # 512 MB streaming operands
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(n):
    with torch.cuda.stream(s1):
        for _ in range(n):
            if BF: _lib.check(lib.buddy_gemm_winograd_domain_bf16x3(P(A), U3.data_ptr(), P(Cm), Mt, N, K, 36, s1.cuda_stream))
            else: _lib.check(lib.buddy_gemm_winograd_domain(P(A), P(Bt), P(Cm), Mt, N, K, 36, s1.cuda_stream))
def mem(n):
    with torch.cuda.stream(s2):
        for _ in range(n):
            torch.mul(x, 1.0001, out=y)
def timed(f):
    f(); torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
ng = 6
tg = timed(lambda: gemm(ng))
tm1 = timed(lambda: mem(1))
nm = max(1, int(round(tg / tm1)))
tm = timed(lambda: mem(nm))
tb = timed(lambda: (gemm(ng), mem(nm)))
print(f"pad={os.environ.get('BUDDY_GEMM_LDS_PAD', '0'):>6}  GEMM x{ng}: {tg:7.2f} ms ({2.0*36*Mt*N*K*ng/tg/1e9:6.1f} TF) | stream kernel x{nm}: {tm:7.2f} ms "
      f"({2*x.numel()*4*nm/tm/1e6:6.0f} GB/s) | both: {tb:7.2f} ms  -> overlap gain {(tg+tm)/tb:.2f}x (ideal {(tg+tm)/max(tg,tm):.2f}x)")
