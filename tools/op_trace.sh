#!/bin/bash
# per-(kernel, grid) average durations of the operator update alone (tools/op_only.py), and the in-order kernel list of one Adam iteration.
# usage: tools/op_trace.sh
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_op -o tr -- python $R/tools/op_only.py 8 4 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tr_op/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
short = lambda n: n.replace("buddy::(anonymous namespace)::", "").replace("void ", "").split("(")[0][-34:]
for r in rows:
    d[(short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{k[0]:36s} grid {k[1]:>7s} {k[2]:>3s} {k[3]:>2s}  x{len(v):4d}  avg {sum(v) / len(v):6.1f} us  total {sum(v) / 1e3:6.2f} ms")
# one iteration, in order: from the last design_dm_kernel to the end
idx = [i for i, r in enumerate(rows) if "design_row_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
print("--- one iteration (start us, duration us, gap to previous end us)")
prev = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f} {((s - prev) / 1e3 if prev else 0):5.1f}  {short(r['Kernel_Name'])}")
    prev = e
print(f"iteration wall {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
PY
rm -rf gpurun_out/tr_op
