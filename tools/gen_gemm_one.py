"""The general GEMM (1x1 convolutions / NIN layers) at the network's shapes IN ISOLATION: buddy_gemm_bf16x3 against buddy_gemm_f16x2, operands rotated through
enough buffers that no launch finds its A rows in the Infinity Cache.  Prints ms and algorithmic GB/s (A read once + C written once) per shape.
usage: [BUDDY_GEN_ROWS=32|64] python tools/gen_gemm_one.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd import _lib

lib = _lib.require_gpu(); P = _lib.ptr; S = _lib.stream_ptr
SHAPES = [(1048576, 128, 384), (1048576, 128, 256), (262144, 256, 512), (262144, 256, 384), (262144, 256, 256), (262144, 128, 256), (262144, 128, 128),
          (262144, 256, 128), (65536, 256, 512), (65536, 256, 256), (16384, 256, 256)]


def pack(W, arith):
    N, K = W.shape
    if arith == "f16x2":
        W3 = torch.empty(lib.buddy_wgemm_f16x2_packed_bytes(1, N, K) // 4, dtype=torch.int32, device="cuda")
        _lib.check(lib.buddy_wgemm_f16x2_pack_weights(P(W), W3.data_ptr(), 1, N, K, S()))
    else:
        W3 = torch.empty(N * K * 6 // 4, dtype=torch.int32, device="cuda")
        _lib.check(lib.buddy_wgemm_pack_weights(P(W), W3.data_ptr(), 1, N, K, S()))
    return W3


def main():
    out = []
    for M, N, K in SHAPES:
        per = 4 * M * (K + N)
        nbuf = max(2, min(16, int(1.5e9 // per) + 1))
        A = [torch.randn(M, K, device="cuda") for _ in range(nbuf)]
        Cc = [torch.empty(M, N, device="cuda") for _ in range(nbuf)]
        W = torch.randn(N, K, device="cuda") / K ** 0.5
        bias = torch.randn(N, device="cuda")
        row = {"M": M, "N": N, "K": K, "nbuf": nbuf}
        for arith in ("bf16x3", "f16x2"):
            W3 = pack(W, arith)
            gemm = lib.buddy_gemm_f16x2 if arith == "f16x2" else lib.buddy_gemm_bf16x3
            run = lambda i: _lib.check(gemm(P(A[i % nbuf]), K, None, 0, 0, W3.data_ptr(), P(Cc[i % nbuf]), N, M, N, K, P(bias), 1.0, 0, S()))
            for i in range(nbuf):
                run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 4 * nbuf
            e0.record()
            for i in range(reps):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row[arith + "_ms"] = ms
            row[arith + "_GBps"] = per / ms / 1e6
        out.append(row)
        print(row, flush=True)
        del A, Cc
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
