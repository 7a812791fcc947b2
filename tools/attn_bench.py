"""Attention kernels in isolation: forward and backward (dq + dkv + delta) time for the shapes of the 4 s (B=8, T=2048) and 30 s (B=4, T=15040)
configurations, C=256, per precision mode.  usage: python tools/attn_bench.py   (BUDDY_ATTN_NW=4|8 forces the fp32 forward tile height)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib
lib = _lib.require_gpu()
P = _lib.ptr; S = _lib.stream_ptr
C = 256
for B, T in ((8, 2048), (4, 15040)):
    q, k, v, dO = (torch.randn(B, T, C, device="cuda") for _ in range(4))
    O = torch.empty_like(q); lse = torch.empty(B, T, device="cuda"); dl = torch.empty(B, T, device="cuda")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    fl = 4.0 * B * T * T * C
    for prec in (0, 1, 2):
        if prec == 0:
            f = lambda: _lib.check(lib.buddy_flash_attention_fwd(P(q), P(k), P(v), P(O), P(lse), B, T, C, C ** -0.5, 0, S()))
            b = lambda: _lib.check(lib.buddy_flash_attention_bwd(P(q), P(k), P(v), P(O), P(dO), P(lse), P(dl), P(dq), P(dk), P(dv), B, T, C, C ** -0.5, 0, S()))
        else:       # 16-bit operands: pre-pass + kernels (csrc/attn16.hip)
            ws = torch.empty(lib.buddy_flash_attention16_workspace(B, T, C), device="cuda")
            f = lambda: _lib.check(lib.buddy_flash_attention16_fwd(P(q), P(k), P(v), P(O), P(lse), B, T, C, C ** -0.5, prec, P(ws), S()))
            b = lambda: _lib.check(lib.buddy_flash_attention16_bwd(P(q), P(k), P(v), P(O), P(dO), P(lse), P(dl), P(dq), P(dk), P(dv), B, T, C, C ** -0.5, prec, P(ws), S()))
        out = []
        for fn in (f, b):
            fn(); torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(3): fn()
            torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 3)
        print(f"B={B} T={T} prec={prec} NW={os.environ.get('BUDDY_ATTN_NW', 'auto')}: fwd {out[0]*1e3:8.3f} ms ({fl/out[0]/1e12:6.1f} TF)   bwd {out[1]*1e3:8.3f} ms ({2.5*fl/out[1]/1e12:6.1f} TF)", flush=True)
