"""The large-tile batched GEMM (csrc/wgemm2.hip, option wgemm_v2) against the production kernel (csrc/wgemm.hip): bit-identical outputs (same products,
same accumulation order) and launch times per shape.  Each variant runs in its own process (the option is a process default for the stand-alone entry).
usage: python tools/wgemm_v2_check.py [Mt N K] ..."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from buddy_amd import _lib
lib = _lib.require_gpu(); P = _lib.ptr; S = _lib.stream_ptr
Mt, N, K, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
nb = 64
g = torch.Generator(device="cpu").manual_seed(Mt + N + K)
A = torch.randn(nb, Mt, K, generator=g).cuda(); Bt = torch.randn(nb, N, K, generator=g).cuda(); Cm = torch.zeros(nb, Mt, N, device="cuda")
U3 = torch.empty(nb * N * K * 6 // 4, dtype=torch.int32, device="cuda")
_lib.check(lib.buddy_wgemm_pack_weights(P(Bt), U3.data_ptr(), nb, N, K, S()))
f = lambda: _lib.check(lib.buddy_gemm_winograd_domain_bf16x3(P(A), U3.data_ptr(), P(Cm), Mt, N, K, nb, S()))
f(); f(); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
ref = torch.einsum("pmk,pnk->pmn", A[:2, :512].double(), Bt[:2].double())
err = float((Cm[:2, :512].double() - ref).abs().max() / ref.abs().max())
np.save(out, Cm[::9, ::7].cpu().numpy())
print(f"large-tile={os.environ.get('BUDDY_WGEMM_V2','0')} Mt={Mt} N={N} K={K}: {dt*1e3:.3f} ms  {12.0*nb*Mt*N*K/dt/1e12:.0f} TF bf16  {(Mt*K+Mt*N)*nb*4/dt/1e9:.0f} GB/s  err vs fp64 {err:.1e}", flush=True)
'''
shapes = [tuple(int(v) for v in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)] or [(29696, 128, 128), (29584, 128, 256), (7568, 256, 256), (7568, 256, 512), (1936, 256, 256)]
import numpy as np
with tempfile.TemporaryDirectory() as d:
    for Mt, N, K in shapes:
        outs = []
        for v in ("0", "1", "0", "1"):
            o = os.path.join(d, f"o{v}.npy")
            r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(Mt), str(N), str(K), o], env=dict(os.environ, BUDDY_WGEMM_large-tile=v), capture_output=True, text=True)
            print(r.stdout.strip() or r.stderr[-400:])
            outs.append(np.load(o))
        print("   bit-identical:", bool(np.array_equal(outs[0], outs[1])))
