#!/bin/bash
# B = 1 (the reference's own shape: one utterance at a time) blind step: un-profiled ms/step, rocprofv3 kernel stats, and the busy / idle split of a step.
# usage: bash tools/b1_profile.sh <tag> [batch]
TAG=${1:-b1}; B=${2:-1}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
FLAGS="--batch $B --steps 20 --warmup 3 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest"
python bench.py $FLAGS > $OUT/${TAG}_bench_B$B.json 2> $OUT/${TAG}_bench_B$B.err; python - <<PY
import json; d = json.loads(open("$OUT/${TAG}_bench_B$B.json").read().strip().splitlines()[-1]); print("B=$B ms_per_step", d["ms_per_step"], "value", d["value"])
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $R/bench.py $FLAGS > $OUT/${TAG}_bench_B${B}_under_rocprof.json 2> $OUT/${TAG}_prof.err
cd $R
find $OUT/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_B${B}_kernel_stats.csv \;
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/prof_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last 15 steps: find the perturb kernel launches as step markers
marks = [i for i, r in enumerate(rows) if "perturb" in r[2]]
print("steps seen", len(marks))
a, b = marks[-16], marks[-1]
seg = rows[a:b]; n = 15
busy = sum(e - s for s, e, _ in seg) / 1e3 / n
span = (rows[b][0] - rows[a][0]) / 1e3 / n
gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:60], seg[i + 1][2][:60]) for i in range(len(seg) - 1))
print("per step: span %.1f us, busy %.1f us, idle %.1f us, launches %d" % (span, busy, span - busy, len(seg) // n))
d = collections.defaultdict(float); c = collections.Counter()
for s, e, k in seg:
    k = k.replace("buddy::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]; d[k] += (e - s) / 1e3 / n; c[k] += 1
for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:40]:
    print("%-50s %5.1f /step  %8.1f us/step  avg %7.1f us" % (k, c[k] / n, v, v * n / c[k]))
print("largest gaps:"); [print("  %.1f us  %s -> %s" % g) for g in gaps[-8:]]
PY
rm -rf $OUT/prof_$TAG
