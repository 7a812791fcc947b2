#!/bin/bash
# How much do the kernels of the two half-batch convolution pipelines (option overlap) actually run side by side?  Kernel trace of a short bench run;
# for the last step: wall span, sum of kernel durations, time with >= 2 kernels in flight, and what runs beside the batched GEMM.
# usage: bash tools/overlap_trace.sh [0|1]   (BUDDY_OVERLAP value, default 1)
export TMPDIR=/tmp; R=$(pwd); cd /tmp
export BUDDY_OVERLAP=${1:-1}
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_ov -o tr -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tr_ov/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("buddy::(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
idx = [i for i, r in enumerate(rows) if "wgemm_bf16x3_kernel<false" in r["Kernel_Name"]]
gem = idx[-1]
# last step = from the last "dps_update"-like boundary: take the last 1/3 of GEMM launches' span
n = len(idx) // 3
seg = rows[idx[-n]: idx[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
ev = []
for r in seg:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur = 0; last = t0; busy1 = busy2 = 0
for t, d in ev:
    if cur >= 1: busy1 += t - last
    if cur >= 2: busy2 += t - last
    cur += d; last = t
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"BUDDY_OVERLAP={__import__('os').environ.get('BUDDY_OVERLAP')}: last step: {len(seg)} kernels, span {(t1 - t0) / 1e6:.2f} ms, sum of durations {tot / 1e6:.2f} ms, "
      f">= 1 kernel in flight {busy1 / 1e6:.2f} ms, >= 2 in flight {busy2 / 1e6:.2f} ms, queues {sorted(set(r.get('Queue_Id', '?') for r in seg))}")
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    a = agg[short(r["Kernel_Name"])]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {k:46s} x{c:4d} {d / 1e6:8.3f} ms  avg {d / c / 1e3:8.1f} us")
PY
rm -rf gpurun_out/tr_ov
