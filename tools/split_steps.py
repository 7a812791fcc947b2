"""Experiment: the blind DPS step for B utterances as S concurrent sub-batches on S HIP streams (each its own network handle / operator).
usage: python tools/split_steps.py [B] [S] [steps]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
a = types.SimpleNamespace(T=50, operator="hip", length=64000)
dev = torch.device("cuda", 0)
runners, streams = [], []
for s in range(S):
    st = torch.cuda.Stream() if S > 1 else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        args, net, edm, tester, seg, y, op = bench.build_stack(a, dev, B // S, s * (B // S))
        runners.append(bench.StepRunner(tester, y, op, dev))
    streams.append(st)
def go(n):
    for _ in range(n):
        for r, st in zip(runners, streams):
            with torch.cuda.stream(st):
                r.step()
go(2); torch.cuda.synchronize(); t = time.perf_counter()
go(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
print(f"B={B} as {S} sub-batches: {dt*1e3:.1f} ms/step  {B/dt:.1f} utterance-steps/s")
