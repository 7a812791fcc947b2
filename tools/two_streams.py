"""Feasibility: two half-batches of the score network (forward + input-VJP) on two HIP streams vs one full batch on one stream.
HBM-bound GroupNorm kernels of one half can overlap the MFMA-bound convolutions of the other.  usage: python tools/two_streams.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_hip_network import build
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = 64000
def loop(nets, xs, cns, cots, streams, reps):
    for _ in range(reps):
        for net, x, cn, cot, st in zip(nets, xs, cns, cots, streams):
            with torch.cuda.stream(st):
                xg = x.requires_grad_(True)
                y = net(xg, cn); g, = torch.autograd.grad(y, xg, cot)
def timed(nets, xs, cns, cots, streams, reps=6):
    loop(nets, xs, cns, cots, streams, 2); torch.cuda.synchronize(); t = time.perf_counter()
    loop(nets, xs, cns, cots, streams, reps); torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
n1 = build(128, 510, 128, 0)
mk = lambda b: ((0.1 * torch.randn(b, L)).cuda(), torch.full((b,), -0.7).cuda(), torch.randn(b, L).cuda())
x, cn, cot = mk(B)
t1 = timed([n1], [x], [cn], [cot], [torch.cuda.current_stream()])
print(f"one stream,  B={B}: {t1*1e3:.1f} ms per fwd+vjp of {B} utterances")
n2 = build(128, 510, 128, 0)
h = B // 2
a, b_ = mk(h), mk(h)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t2 = timed([n1, n2], [a[0], b_[0]], [a[1], b_[1]], [a[2], b_[2]], [s1, s2])
print(f"two streams, 2 x B={h}: {t2*1e3:.1f} ms per fwd+vjp of {B} utterances  ({t1/t2:.3f}x)")
t3 = timed([n1, n2], [a[0], b_[0]], [a[1], b_[1]], [a[2], b_[2]], [s1, s1])
print(f"one stream,  2 x B={h} back to back: {t3*1e3:.1f} ms")
