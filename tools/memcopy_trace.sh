#!/bin/bash
# Which HIP API calls does ONE sampler step issue from the host?  hip-runtime + kernel + memory-copy trace of a short bench run; the host-side step
# boundaries are the launches of the sampler's perturb kernel (correlation ids tie kernels to API records).  Output: gpurun_out/<tag>_api_per_step.txt
TAG=${1:-r06}
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --memory-copy-trace --kernel-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/mc -o mc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2>&1
cd $R
python - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
kt = list(csv.DictReader(open(glob.glob("gpurun_out/mc/**/*kernel_trace.csv", recursive=True)[0])))
api = list(csv.DictReader(open(glob.glob("gpurun_out/mc/**/*hip_api_trace.csv", recursive=True)[0])))
mc = list(csv.DictReader(open(glob.glob("gpurun_out/mc/**/*memory_copy_trace.csv", recursive=True)[0])))
api.sort(key=lambda r: int(r["Start_Timestamp"]))
by_corr = {r["Correlation_Id"]: r for r in api}
marks = sorted(int(by_corr[k["Correlation_Id"]]["Start_Timestamp"]) for k in kt if "perturb_kernel" in k["Kernel_Name"] and k["Correlation_Id"] in by_corr)
out = open(f"gpurun_out/{tag}_api_per_step.txt", "w")
def P(*a):
    print(*a); print(*a, file=out)
P("steps found (perturb launches):", len(marks))
kern_by_corr = {k["Correlation_Id"]: k for k in kt}
for s in range(len(marks) - 1):
    lo, hi = marks[s], marks[s + 1]
    c = collections.Counter(); t = collections.Counter()
    for r in api:
        st = int(r["Start_Timestamp"])
        if lo <= st < hi:
            c[r["Function"]] += 1; t[r["Function"]] += int(r["End_Timestamp"]) - st
    P(f"--- step {s}: host interval {(hi - lo) / 1e6:.2f} ms")
    for k, v in c.most_common(14):
        P(f"   {v:6d} x {k:32s} host time {t[k] / 1e6:8.3f} ms")
    # the synchronous copies: when (relative to the step start) and how long the host sat in them
    for r in api:
        st = int(r["Start_Timestamp"])
        if lo <= st < hi and r["Function"] in ("hipMemcpyWithStream", "hipMemcpy", "hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize"):
            P(f"      {r['Function']:24s} at +{(st - lo) / 1e6:8.3f} ms, {(int(r['End_Timestamp']) - st) / 1e3:9.1f} us")
    dirs = collections.Counter()
    for r in mc:
        a = by_corr.get(r["Correlation_Id"])
        if a and lo <= int(a["Start_Timestamp"]) < hi:
            dirs[(r["Direction"], a["Function"])] += 1
    P("   memory copies:", dict(dirs))
    blit = collections.Counter()
    for k in kt:
        a = by_corr.get(k["Correlation_Id"])
        if a and lo <= int(a["Start_Timestamp"]) < hi and ("copyBuffer" in k["Kernel_Name"] or "fillBuffer" in k["Kernel_Name"]):
            blit[(k["Kernel_Name"][:28], a["Function"], k.get("Grid_Size_X", k.get("Grid_Size")))] += 1
    P("   blit kernels by issuing API:", dict(blit))
PY
rm -rf gpurun_out/mc
