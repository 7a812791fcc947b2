import sys, torch
sys.path.insert(0, ".")
from buddy_amd import _lib
lib = _lib.require_gpu()
n = 2 * 2 ** 30 // 4
x = torch.randn(n, device="cuda"); y = torch.empty(1024, device="cuda")
S = torch.cuda.current_stream().cuda_stream
def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
for rb in (512, 1024, 2048):
    for mode in (3, 4):
        for lds in (1, 32, 48, 64):       # KB of dynamic LDS per workgroup: 1 = unbounded occupancy, 32 / 48 / 64 = at most 5 / 3 / 2 workgroups per CU
            dt = t(lambda: _lib.check(lib.buddy_hbm_ubench(x.data_ptr(), y.data_ptr(), n * 4, mode, rb, lds, S)))
            print(f"rows of {rb} B, {'all stages in flight' if mode == 3 else 'staged'}, lds {lds} KB: {n * 4 / dt / 1e9:.0f} GB/s")
for nt in (0, 1):
    dt = t(lambda: _lib.check(lib.buddy_hbm_ubench(x.data_ptr(), y.data_ptr(), n * 4, 1, nt, -4, S)))
    print(f"contiguous read nt={nt}: {n * 4 / dt / 1e9:.0f} GB/s")
