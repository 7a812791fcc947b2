"""fp64 arbiter on the GPU vs on the CPU (same torch ops, float64): per-step SI-SDR between the two trajectories for the small configuration
(CPU trace from oracle/_ref/arbiter), then the time of a few full-size float64 steps on the GPU.  usage: python tools/fp64_gpu_check.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.arbiter_runs import run_blind
from buddy_amd.utils.metrics import si_sdr
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "arbiter", "L8192_T10_nf32_up10")
sd = lambda a, b: round(float(si_sdr(torch.as_tensor(a)[None].double(), torch.as_tensor(b)[None].double())), 1)
for s in (0, 1):
    cpu = np.load(os.path.join(d, f"seed{s}_fp64.npz"))["xden"]
    t0 = time.time(); g, _, _ = run_blind(s, 8192, 10, 32, 10, 2000, fp64=True, device="cuda"); dt = time.time() - t0
    print(f"seed {s}: fp64 GPU vs fp64 CPU per step (float32-stored traces, floor ~140 dB):", [sd(g[i], cpu[i]) for i in range(10)], f"{dt:.1f} s", flush=True)
t0 = time.time(); g, _, _ = run_blind(0, 64000, 3, 128, 10, 8000, fp64=True, device="cuda"); torch.cuda.synchronize()
print(f"full size (L=64000, nf=128): 3 fp64 steps on the GPU in {time.time() - t0:.1f} s; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
