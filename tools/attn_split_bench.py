"""fp32 attention kernels with their sequential loop split over 1 ... 16 workgroups per row block (buddy_flash_attention_*_split): forward and backward
time at T = 2048 (one 4 s utterance), C = 256, for B = 1, 2, 4, 8.  usage: python tools/attn_split_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
P = _lib.ptr; S = _lib.stream_ptr
C, T = 256, 2048
for B in (1, 2, 4, 8):
    q, k, v, dO = (torch.randn(B, T, C, device="cuda") for _ in range(4))
    O = torch.empty_like(q); lse = torch.empty(B, T, device="cuda"); dl = torch.empty(B, T, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    print(f"B={B}: picked splits = {lib.buddy_flash_attention_splits(B, T)}")
    for ns in (1, 2, 3, 4, 6, 8, 11, 16, 22, 32):
        nb = (T + 31) // 32
        if -(-nb // ns) * (ns - 1) >= nb: continue
        ws = torch.empty(max(1, lib.buddy_flash_attention_workspace(B, T, C, ns)), device="cuda")
        f = lambda: _lib.check(lib.buddy_flash_attention_fwd_split(P(q), P(k), P(v), P(O), P(lse), B, T, C, C ** -0.5, ns, P(ws), S()))
        b = lambda: _lib.check(lib.buddy_flash_attention_bwd_split(P(q), P(k), P(v), P(O), P(dO), P(lse), P(dl), P(dq), P(dk), P(dv), B, T, C, C ** -0.5, ns, P(ws), S()))
        out = []
        for fn in (f, b):
            fn(); torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 10)
        print(f"  splits={ns:2d}: fwd {out[0]*1e3:7.3f} ms   bwd {out[1]*1e3:7.3f} ms", flush=True)
