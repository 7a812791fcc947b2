#!/bin/bash
# rocprofv3 kernel stats of a short bench run under one environment setting: all kernels matching a pattern + the total.
# usage: bash tools/prof_env.sh VAR=VALUE 'regex' [bench flags]
KV=$1; PAT=$2; shift 2
R=$(pwd); OUT=/tmp/prof_env_$$; export TMPDIR=/tmp
cd /tmp
env $KV rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest $@ > /dev/null 2>&1
cd $R
python - "$(find $OUT -name '*kernel_stats.csv')" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1]))); pat = re.compile(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms %.1f" % (tot / 1e6))
for r in rows:
    n = r["Name"].replace("buddy::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if pat.search(n): print("%-52s %5d %9.2f ms  avg %8.1f us" % (n[:52], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf $OUT
