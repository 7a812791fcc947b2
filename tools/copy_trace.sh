#!/bin/bash
# who issues the device-to-device copies of a step?  kernel trace of a short bench run; every __amd_rocclr_copyBuffer with its duration and neighbours.
TAG=${1:-cp}; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$TAG -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/tr_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows); pat = collections.Counter(); dur = collections.defaultdict(float)
for i, r in enumerate(rows):
    if "copyBuffer" not in r["Kernel_Name"]: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    short = lambda k: k["Kernel_Name"].replace("buddy::(anonymous namespace)::", "").replace("void ", "")[:48]
    key = (short(rows[i - 1]) if i else "-", short(rows[i + 1]) if i + 1 < n else "-", r.get("Grid_Size", r.get("Grid_Size_X", "")))
    pat[key] += 1; dur[key] += d
print("kernels", n, "copies", sum(pat.values()))
for k, c in pat.most_common(25): print(f"{c:5d} x avg {dur[k]/c:7.1f} us  grid {k[2]:>10s}  after [{k[0]}] before [{k[1]}]")
PY
rm -rf $OUT/tr_$TAG
