#!/bin/bash
# per-(kernel, grid) average durations of one bench run (kernel trace).  usage: [BENCH_FLAGS="--batch 1"] tools/trace_by_grid.sh <name-substring> ...
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr -o tr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest $BENCH_FLAGS > /dev/null 2>&1
cd $R
python - "$@" <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/tr/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in sys.argv[1:]):
        key = (n.replace("buddy::(anonymous namespace)::", "").split("(")[0][-40:], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
        d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), round(sum(v) / len(v), 1), "us", "%.1f%%" % (100 * sum(v) / tot))
PY
rm -rf gpurun_out/tr
