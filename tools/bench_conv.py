"""Micro-benchmarks on the MI355X: pure fp32-MFMA issue rate (sustained matrix peak at the DVFS clock) and the 3x3
implicit-GEMM conv at the network's shapes.  python tools/bench_conv.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
S = lambda: torch.cuda.current_stream().cuda_stream

def ubench(mode):
    seed = {"zero": torch.zeros(1024), "const": torch.full((1024,), 1.5), "rand": torch.randn(1024)}[mode].cuda()
    blocks, iters = 256 * 3, 20000
    out = torch.empty(blocks * 256, device="cuda"); clk = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(2):
        _lib.check(lib.buddy_mfma_ubench(seed.data_ptr(), out.data_ptr(), blocks, iters, clk.data_ptr(), S()))
    torch.cuda.synchronize(); t = time.perf_counter()
    _lib.check(lib.buddy_mfma_ubench(seed.data_ptr(), out.data_ptr(), blocks, iters, clk.data_ptr(), S()))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    fl = blocks * 4 * 4 * iters * 4096.0
    c = clk.cpu().tolist()
    print(f"mfma ubench [{mode:5s}]: {fl/dt/1e12:7.1f} TFLOP/s   shader clock {c[0]/(c[1]/100e6)/1e9:.3f} GHz  ({dt*1e3:.1f} ms)")

def conv(B, H, W, Cin, Cout, reps=5):
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, 9 * Cin, device="cuda") / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
    for _ in range(2):
        _lib.check(lib.buddy_conv3x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        _lib.check(lib.buddy_conv3x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"conv3x3 B{B} {H}x{W} {Cin:3d}->{Cout:3d}: {dt*1e3:8.3f} ms  {fl/dt/1e12:6.1f} TFLOP/s")

def conv_wino(B, H, W, Cin, Cout, reps=5):
    import numpy as np
    x = torch.randn(B, H, W, Cin, device="cuda"); w = (torch.randn(Cout, 9 * Cin) / (9 * Cin) ** 0.5).numpy()
    U = np.empty(16 * Cin * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd_transform_weights(w.ctypes.data, Cout, Cin, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda()
    b = torch.randn(Cout, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
    for _ in range(2):
        _lib.check(lib.buddy_conv3x3_winograd(x.data_ptr(), Ud.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        _lib.check(lib.buddy_conv3x3_winograd(x.data_ptr(), Ud.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"winograd B{B} {H}x{W} {Cin:3d}->{Cout:3d}: {dt*1e3:8.3f} ms  {fl/dt/1e12:6.1f} TFLOP/s (direct-conv algorithmic flops; executed {fl*4/9/dt/1e12:5.1f})")


if __name__ == "__main__":
    for m in ["zero", "const", "rand"]:
        ubench(m)
    for shp in [(8, 512, 256, 256, 256), (8, 512, 256, 128, 128), (8, 512, 256, 384, 128), (8, 256, 128, 512, 256), (8, 128, 64, 256, 256),
                (8, 64, 32, 512, 256), (1, 512, 256, 256, 256), (1, 64, 32, 512, 256)]:
        conv(*shp)
        conv_wino(*shp)
