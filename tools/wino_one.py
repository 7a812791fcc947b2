import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from buddy_amd import _lib
lib = _lib.require_gpu()
S = lambda: torch.cuda.current_stream().cuda_stream
def run(B, H, W, Cin, Cout, reps=5):
    x = torch.randn(B, H, W, Cin, device="cuda"); w = (torch.randn(Cout, 9 * Cin) / (9 * Cin) ** 0.5).numpy()
    U = np.empty(16 * Cin * Cout, dtype=np.float32)
    _lib.check(lib.buddy_winograd_transform_weights(w.ctypes.data, Cout, Cin, U.ctypes.data))
    Ud = torch.from_numpy(U).cuda(); b = torch.randn(Cout, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
    for _ in range(2):
        _lib.check(lib.buddy_conv3x3_winograd(x.data_ptr(), Ud.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        _lib.check(lib.buddy_conv3x3_winograd(x.data_ptr(), Ud.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, S()))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / reps
    print(f"wino B{B} {H}x{W} {Cin}->{Cout}: {dt*1e3:.3f} ms  eff {2.0*B*H*W*Cout*9*Cin/dt/1e12:.1f} TF")
for B in (1, 2, 4, 8):
    run(B, 512, 256, 256, 256)
run(1, 512, 256, 256, 32); run(1, 64, 32, 256, 256); run(8, 512, 256, 32, 32)
