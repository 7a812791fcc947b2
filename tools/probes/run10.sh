cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_cp -o tr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2>&1
cd $R
f=$(find gpurun_out/tr_cp -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f 12 | tail -8
python - <<PY
import csv, collections
rows = sorted(csv.DictReader(open("$f")), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("buddy::(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
# last step only: after the last-but-one dps_update-like kernel; use the final 1/6 of rows
seg = rows[-len(rows)//6:]
c = collections.Counter(); nxt = collections.Counter()
for i, r in enumerate(seg):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = short(seg[i-1]["Kernel_Name"]) if i else "?"
        nx = short(seg[i+1]["Kernel_Name"]) if i + 1 < len(seg) else "?"
        c[(prev, nx, r["Grid_Size_X"], r.get("Workgroup_Size_X", ""))] += 1
print("copyBuffer in the last sixth of the trace:", sum(c.values()))
for k, v in c.most_common(25): print(v, k)
PY
rm -rf gpurun_out/tr_cp
