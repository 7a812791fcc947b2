"""F(4x4,3x3) three-pass conv: correctness vs torch fp64 and timing vs the fused F(2x2,3x3) kernel.  usage: python tools/wino4_one.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from buddy_amd import _lib
lib = _lib.require_gpu()
S = lambda: torch.cuda.current_stream().cuda_stream
def check():
    B, H, W, Cin, Cout = 2, 32, 24, 16, 12
    torch.manual_seed(0)
    x = torch.randn(B, H, W, Cin, device="cuda"); wt = torch.randn(Cout, 3, 3, Cin) / (9 * Cin) ** 0.5
    w = wt.reshape(Cout, 9 * Cin).numpy().copy(); U = np.empty(36 * Cin * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd4_transform_weights(w.ctypes.data, Cout, Cin, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda(); b = torch.randn(Cout, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
    sc = torch.empty(36 * (B * H * W // 16) * (Cin + Cout), device="cuda")
    _lib.check(lib.buddy_conv3x3_winograd4(x.data_ptr(), Ud.data_ptr(), b.data_ptr(), y.data_ptr(), sc.data_ptr(), B, H, W, Cin, Cout, S()))
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double().cuda(), b.double(), padding=1).permute(0, 2, 3, 1)
    print("F(4,3) max err", float((y - ref).abs().max()), "ref max", float(ref.abs().max()))
def run(B, H, W, Cin, Cout, reps=5):
    x = torch.randn(B, H, W, Cin, device="cuda"); w = (torch.randn(Cout, 9 * Cin) / (9 * Cin) ** 0.5).numpy()
    U = np.empty(36 * Cin * Cout, dtype=np.float32); U2 = np.empty(16 * Cin * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd4_transform_weights(w.ctypes.data, Cout, Cin, U.ctypes.data))
    _lib.check(lib.buddy_winograd_transform_weights(w.ctypes.data, Cout, Cin, U2.ctypes.data))
    Ud = torch.from_numpy(U).cuda(); U2d = torch.from_numpy(U2).cuda(); b = torch.randn(Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda"); y2 = torch.empty_like(y)
    sc = torch.empty(36 * (B * H * W // 16) * (Cin + Cout), device="cuda")
    f4 = lambda: _lib.check(lib.buddy_conv3x3_winograd4(x.data_ptr(), Ud.data_ptr(), b.data_ptr(), y.data_ptr(), sc.data_ptr(), B, H, W, Cin, Cout, S()))
    f2 = lambda: _lib.check(lib.buddy_conv3x3_winograd(x.data_ptr(), U2d.data_ptr(), b.data_ptr(), y2.data_ptr(), B, H, W, Cin, Cout, S()))
    out = []
    for f in (f4, f2):
        f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / reps)
    fl = 2.0 * B * H * W * Cout * 9 * Cin
    print(f"B{B} {H}x{W} {Cin}->{Cout}: F(4,3) {out[0]*1e3:.3f} ms ({fl/out[0]/1e12:.0f} TF)  F(2,3) fused {out[1]*1e3:.3f} ms ({fl/out[1]/1e12:.0f} TF)  "
          f"diff {float((y - y2).abs().max() / y2.abs().max()):.1e}")
check()
for s in ((8, 512, 256, 256, 256), (8, 512, 256, 128, 128), (8, 512, 256, 384, 128), (8, 512, 256, 128, 384), (8, 256, 128, 256, 256), (8, 256, 128, 512, 256),
          (8, 128, 64, 256, 256), (8, 64, 32, 256, 256), (1, 512, 256, 256, 256)):
    run(*s)
