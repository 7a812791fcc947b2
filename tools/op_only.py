"""Only the blind-operator update (optimize_op, 10 Adam iterations) for U utterances of 4 s -- for profiling that loop in isolation.
usage: python tools/op_only.py [U] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd.config import compose
from buddy_amd.testing.operators.subband_filtering import BlindSubbandFiltering
U = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = 64000
args = compose(tester="blind_dereverberation_BUDDy")
op = BlindSubbandFiltering(args.tester.informed_dereverberation.op_hp, sample_rate=16000, num_utts=U, device="cuda", length=L)
op.update_H(use_noise=True)
torch.manual_seed(0)
y = 0.05 * torch.randn(U, L, device="cuda"); x = 0.05 * torch.randn(U, L, device="cuda")
op.hip_bind(y, args.tester.posterior_sampling)
for _ in range(2): op.hip_optimize(x, 0.3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): op.hip_optimize(x, 0.3)
torch.cuda.synchronize()
print(f"optimize_op U={U}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per call (10 iterations)")
