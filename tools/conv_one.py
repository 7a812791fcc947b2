"""One conv shape, a few launches (for rocprofv3 --pmc passes).  python tools/conv_one.py B H W Cin Cout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
B, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, 9 * Cin, device="cuda") / (9 * Cin) ** 0.5
b = torch.randn(Cout, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
for _ in range(3):
    _lib.check(lib.buddy_conv3x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("algorithmic bytes (read x once + weights once + write y once):", (x.numel() + w.numel() + y.numel()) * 4)
