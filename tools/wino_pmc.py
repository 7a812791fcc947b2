import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from buddy_amd import _lib
lib = _lib.require_gpu()
B, H, W, Cin, Cout = 8, 512, 256, 256, 256
x = torch.randn(B, H, W, Cin, device="cuda"); w = (torch.randn(Cout, 9 * Cin) / (9 * Cin) ** 0.5).numpy()
U = np.empty(16 * Cin * Cout, dtype=np.float32)
_lib.check(lib.buddy_winograd_transform_weights(w.ctypes.data, Cout, Cin, U.ctypes.data))
Ud = torch.from_numpy(U).cuda(); b = torch.randn(Cout, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
for _ in range(3):
    _lib.check(lib.buddy_conv3x3_winograd(x.data_ptr(), Ud.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
