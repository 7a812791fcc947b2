#!/bin/bash
# per-(kernel, grid) table of EVERY kernel of one bench run (kernel trace), written to gpurun_out/<tag>_by_grid.txt.
# usage: [BENCH_FLAGS="--batch 1"] tools/trace_all.sh <tag>
TAG=${1:-trace}
export TMPDIR=/tmp; R=$(pwd); mkdir -p $R/gpurun_out; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$TAG -o tr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest $BENCH_FLAGS > $R/gpurun_out/${TAG}_trace_bench.json 2> $R/gpurun_out/${TAG}_trace.err
cd $R
python - $TAG <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/tr_{tag}/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("buddy::(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:70]
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("Workgroup_Size_X", ""))
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
with open(f"gpurun_out/{tag}_by_grid.txt", "w") as o:
    o.write(f"total kernel time {tot / 1e3:.2f} ms over the run (6 steps: 1 warm-up + 3 timed + 2 attribution)\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        o.write(f"{k[0]:70s} grid {k[1]:>9s} {k[2]:>5s} {k[3]:>3s} wg {k[4]:>4s}  n {len(v):5d}  avg {sum(v) / len(v):9.1f} us  total {sum(v) / 1e3:8.2f} ms  {100 * sum(v) / tot:5.1f}%\n")
PY
rm -rf gpurun_out/tr_$TAG
head -30 gpurun_out/${TAG}_by_grid.txt
