#!/usr/bin/env python3
"""Per-shape timing of the Winograd-domain batched GEMM (64 positions) in bf16x3 and f16x2 arithmetic on the shipped network's layer shapes at B = 8 x 4 s:
ms, executed MFMA TFLOP/s, algorithmic GB/s (V read + M written once).  usage: python tools/wgemm_modes_bench.py [out.json]"""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from buddy_amd import _lib
lib = _lib.load()
P = lambda t: _lib.fptr(t) if hasattr(_lib, "fptr") else t.data_ptr()
S = lambda: torch.cuda.current_stream().cuda_stream
import ctypes as C
def fp(t): return C.cast(t.data_ptr(), C.POINTER(C.c_float))
shapes = [("level0 128->128", 8, 86 * 43, 128, 128), ("level0 256->128 (skip concat)", 8, 86 * 43, 256, 128), ("level1 256->256", 8, 43 * 22, 256, 256),
          ("level1 512->256", 8, 43 * 22, 512, 256), ("level1 128->256", 8, 43 * 22, 128, 256), ("level2 256->256", 8, 22 * 11, 256, 256), ("level2 512->256", 8, 22 * 11, 512, 256)]
out = []
for name, B, tpu, Cin, Cout in shapes:
    tiles = B * tpu
    V = torch.randn(64, tiles, Cin, device="cuda"); U = torch.randn(64, Cout, Cin, device="cuda") * 0.05
    M = torch.empty(64, tiles, Cout, device="cuda")
    U3 = torch.empty(int(lib.buddy_wgemm_packed_bytes(64, Cout, Cin)) // 4, dtype=torch.int32, device="cuda")
    U2 = torch.empty(int(lib.buddy_wgemm_f16x2_packed_bytes(64, Cout, Cin)) // 4, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_wgemm_pack_weights(fp(U), U3.data_ptr(), 64, Cout, Cin, S()))
    _lib.check(lib.buddy_wgemm_f16x2_pack_weights(fp(U), U2.data_ptr(), 64, Cout, Cin, S()))
    vmax = torch.empty(B, 64, 32, dtype=torch.int32, device="cuda")
    _lib.check(lib.buddy_abs_max_bits(fp(V), 64, B, tpu * Cin, vmax.data_ptr(), S()))
    def t(fn, n=10):
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t3 = t(lambda: _lib.check(lib.buddy_gemm_winograd_domain_bf16x3(fp(V), U3.data_ptr(), fp(M), tiles, Cout, Cin, 64, S())))
    t2 = t(lambda: _lib.check(lib.buddy_gemm_winograd_domain_f16x2(fp(V), U2.data_ptr(), fp(M), tiles, Cout, Cin, 64, vmax.data_ptr(), tpu, S())))
    fl = 2.0 * 64 * tiles * Cin * Cout; by = 4.0 * 64 * tiles * (Cin + Cout)
    r = {"shape": name, "tiles": tiles, "Cin": Cin, "Cout": Cout, "bf16x3_ms": t3, "f16x2_ms": t2, "bf16x3_exec_tflops": 6 * fl / t3 / 1e9, "f16x2_exec_tflops": 3 * fl / t2 / 1e9,
         "bf16x3_GBps": by / t3 / 1e6, "f16x2_GBps": by / t2 / 1e6}
    out.append(r)
    print(f"{name:32s} bf16x3 {t3:.3f} ms ({r['bf16x3_exec_tflops']:.0f} TF, {r['bf16x3_GBps']:.0f} GB/s)   f16x2 {t2:.3f} ms ({r['f16x2_exec_tflops']:.0f} TF, {r['f16x2_GBps']:.0f} GB/s)")
if len(sys.argv) > 1: json.dump(out, open(sys.argv[1], "w"), indent=1)
