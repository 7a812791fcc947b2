"""Per-shape timing of every matrix-core launch in one blind DPS step (or one network fwd+VJP): run with BUDDY_PROF_DUMP=<file>.
usage: BUDDY_PROF_DUMP=/tmp/shapes.txt python tools/gemm_shapes.py [B]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
from tests.test_hip_network import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = 64000
net = build(128, 510, 128, 0)
x = (0.1 * torch.randn(B, L)).cuda().requires_grad_(True); cn = torch.full((B,), -0.7).cuda(); cot = torch.randn(B, L).cuda()
for _ in range(2):
    y = net(x, cn); g, = torch.autograd.grad(y, x, cot)
torch.cuda.synchronize()
lib = _lib.load()
lib.buddy_prof_enable(2)
y = net(x, cn); g, = torch.autograd.grad(y, x, cot)
torch.cuda.synchronize()
import ctypes as C
a = [(C.c_double * 2)() for _ in range(2)]; n = (C.c_longlong * 2)(); b = [(C.c_double * 2)() for _ in range(2)]
lib.buddy_prof_collect(a[0], a[1], n, b[0], b[1])
rows = collections.defaultdict(lambda: [0, 0.0])
for ln in open(os.environ["BUDDY_PROF_DUMP"]):
    k, taps, M, N, K, bt, ms = ln.split()
    r = rows[(int(k), int(taps), int(M), int(N), int(K), int(bt))]; r[0] += 1; r[1] += float(ms)
tot = sum(r[1] for r in rows.values())
print(f"total matrix-core ms {tot:.2f}")
print("kind taps       M     N     K batch calls   ms_tot  ms_avg   TF/s  GB/s(alg)")
for key, (c, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    k, taps, M, N, K, bt = key
    fl = 2.0 * M * N * K * taps * bt * c; by = 4.0 * bt * (M * K + N * K * taps + M * N) * c
    print(f"{k:4d} {taps:4d} {M:7d} {N:5d} {K:5d} {bt:5d} {c:5d} {ms:8.3f} {ms/c:7.3f} {fl/ms/1e9:6.1f} {by/ms/1e6:8.0f}")
