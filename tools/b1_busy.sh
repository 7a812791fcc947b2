#!/bin/bash
# GPU-busy share of the B = 1 blind step: rocprofv3 kernel stats of `bench.py --batch 1` against its own ms_per_step.  usage: tools/b1_busy.sh <tag>
TAG=${1:-b1}; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $R/bench.py --batch 1 --steps 10 --warmup 2 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > $OUT/${TAG}_b1.json 2> /dev/null
cd $R
find $OUT/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_b1_kernel_stats.csv \;
python - <<PY
import csv, json
d = json.loads([l for l in open("$OUT/${TAG}_b1.json") if l.startswith("{")][-1])
rows = list(csv.DictReader(open("$OUT/${TAG}_b1_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) * 1e-6; calls = sum(int(r["Calls"]) for r in rows)
print("ms_per_step", d["ms_per_step"], "total kernel ms (all steps incl. warmup/attribution)", tot, "launches", calls)
for r in rows[:22]: print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])*1e-6:9.2f} ms avg {float(r["AverageNs"])*1e-3:8.1f} us')
PY
rm -rf $OUT/prof_$TAG
