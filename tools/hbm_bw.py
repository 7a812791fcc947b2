"""HBM streaming calibration on this box: the library's own float4 kernels (buddy_hbm_ubench: copy / read / write, plain and non-temporal, over a grid
sweep) next to torch's copy / sum / add kernels.  usage: python tools/hbm_bw.py [out.json]

MI355X_MICROARCH.md records 6.29 TB/s for a float4 copy (8.0 TB/s nominal): the best copy rate found here is the `calibrated` denominator bench.py
quotes beside the nominal one."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buddy_amd import _lib


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    lib = _lib.require_gpu()
    out = {"ubench": [], "torch": {}}
    S = lambda: torch.cuda.current_stream().cuda_stream
    for mb in (1024, 4096):
        n = mb * 2 ** 20 // 4
        x = torch.randn(n, device="cuda")
        y = torch.empty_like(x)
        for mode, name, moved in ((0, "copy", 2), (1, "read", 1), (2, "write", 1)):
            for nt in (0, 1):
                for blocks in (-1, -2, -4, -8, 256 * 2, 256 * 8, 256 * 32, 256 * 64, 256 * 256):
                    dt = timed(lambda: _lib.check(lib.buddy_hbm_ubench(x.data_ptr(), y.data_ptr(), n * 4, mode, nt, blocks, S())))
                    r = {"MB": mb, "mode": name, "nt": nt, "blocks": blocks, "GBps": moved * n * 4 / dt / 1e9}
                    out["ubench"].append(r)
                    print(r, flush=True)
        if mb == 1024:
            out["torch"] = {"copy_GBps": 8 * n / timed(lambda: y.copy_(x)) / 1e9, "add_GBps": 8 * n / timed(lambda: torch.add(x, 1.0, out=y)) / 1e9,
                            "sum_read_GBps": 4 * n / timed(lambda: x.sum()) / 1e9,
                            "memcpy_d2d_GBps": 8 * n / timed(lambda: _lib.check(lib.buddy_copy_d2d(y.data_ptr(), x.data_ptr(), n * 4, S()))) / 1e9}
            print(out["torch"], flush=True)
        del x, y
    best = {m: max((r for r in out["ubench"] if r["mode"] == m), key=lambda r: r["GBps"]) for m in ("copy", "read", "write")}
    out["best"] = best
    print(json.dumps(best))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
