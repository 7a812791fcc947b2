"""Summarise rocprofv3 --pmc passes of bench.py into per-convolution HBM traffic of the dominant kernel group.
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
A 3x3 convolution is three launches (w6_input_kernel / w4_input_kernel, the 64- / 36-batch igemm_kernel<...,36>, w6_output_kernel /
w4_output_kernel); their counters are summed and divided by the number of convolutions (= input-transform launches).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled here."""
import csv, hashlib, json, os, sys

GROUP = ("input transform (w6_input_kernel, w4_input_kernel)", "batched GEMM (wgemm_f16x2_kernel / wgemm_bf16x3_kernel / igemm_kernel<...,36>)",
         "output transform (w6_output_kernel, w4_output_kernel)")
MATCH = {GROUP[0]: ("w6_input_kernel", "w4_input_kernel"), GROUP[1]: ("2, 2, 36>", "wgemm_bf16x3_kernel<false", "wgemm_f16x2_rt2_kernel", "wgemm_f16x2_kernel<"), GROUP[2]: ("w6_output_kernel", "w4_output_kernel")}   # not the <..., true, ...> instantiation: the 1x1 convolutions


def per_kernel(path, counter):
    tot = {k: [0.0, 0] for k in GROUP}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for k in GROUP:
            if any(m in r["Kernel_Name"] for m in MATCH[k]):
                tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
    return tot


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
nconv = f[GROUP[0]][1]
out = {"kernel_group": "3x3 convolution = input transform + batched Winograd-domain GEMM (64 / 36 positions: F(6x6,3x3) / F(4x4,3x3); wgemm_bf16x3_kernel, or "
                       "igemm_kernel<1,false,false,2,2,36> with BUDDY_GEMM=fp32) + output transform",
       "convolutions_in_fetch_pass": nconv, "convolutions_in_write_pass": w[GROUP[0]][1],
       "per_kernel_bytes_per_convolution": {k: {"fetch": 2 * f[k][0] * 1024 / nconv, "write": w[k][0] * 1024 / max(1, w[GROUP[0]][1])} for k in GROUP},
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-concurrent 0`; FETCH_SIZE x2 (gfx950 correction)"}
_h = hashlib.sha1()
for _f in ("igemm.hip", "wgemm.hip", "wino4.hip", "wino6.hip", "common.h"):        # same stamp as bench.py conv_source_stamp(): the summary is tied to these kernels
    _h.update(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "buddy_amd", "csrc", _f), "rb").read())
out["source_stamp"] = _h.hexdigest()[:12]
out["fetch_bytes_per_launch"] = sum(v["fetch"] for v in out["per_kernel_bytes_per_convolution"].values())
out["write_bytes_per_launch"] = sum(v["write"] for v in out["per_kernel_bytes_per_convolution"].values())
out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]


def whole_run(path, counter):
    """every kernel of the run except the peak micro-benchmarks and the runtime's copy / fill kernels (set-up): counter sum, busy time, dispatches"""
    tot, dur, seen = 0.0, 0.0, set()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] != counter or "ubench" in n or "__amd_rocclr" in n or "pack" in n or "prep" in n:
            continue
        tot += float(r["Counter_Value"])
        if r.get("Dispatch_Id") not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            seen.add(r.get("Dispatch_Id")); dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return tot * 1024, dur * 1e-9, len(seen)


fb, fd, fn = whole_run(sys.argv[1], "FETCH_SIZE")
wb, wd, wn = whole_run(sys.argv[2], "WRITE_SIZE")
out["whole_run"] = {"fetch_bytes_x2": 2 * fb, "write_bytes": wb, "kernel_busy_s": (fd + wd) / 2, "dispatches": fn,
                    "avg_hbm_GBps_while_busy": (2 * fb + wb) / ((fd + wd) / 2) / 1e9 if fd + wd > 0 else None,
                    "note": "ALL kernels of the profiled run (sampler steps incl. warm-up and the instrumented attribution steps; without the peak micro-benchmarks, "
                            "weight preparation and the runtime's copy / fill kernels): HBM bytes moved / summed kernel durations = the average HBM rate of the "
                            "step while the GPU is busy (busy ~ wall: the step has no launch gaps).  FETCH_SIZE x2 is the wide-read correction; kernels with narrow "
                            "reads are over-counted by it, so this is an upper estimate"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out)[:900])
