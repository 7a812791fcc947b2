#!/bin/bash
# kernel count, busy time, idle gaps and the share of short kernels in one bench step (kernel trace of `bench.py --steps 3`).  usage: tools/step_gaps.sh
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_gap -o tr -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/tr_gap/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("buddy::(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
# the last step: between the last two design_row_kernel of iteration 0 of an optimize_op ... simpler: last 1/3 of the kernels after the final ubench
idx = [i for i, r in enumerate(rows) if "wgemm_bf16x3_kernel<false" in r["Kernel_Name"] or "wgemm_bf16x3_kernel<3" in r["Kernel_Name"]]
n_per_step = 80
last = idx[-n_per_step:]                       # the 80 Winograd GEMMs of the last step
a = last[0]; b = last[-1]
seg = rows[a:b + 1]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg)
gaps = [(int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3 for i in range(len(seg) - 1)]
print(f"network part of the last step (first to last Winograd GEMM): {len(seg)} kernels, span {span / 1e3:.2f} ms, busy {busy / 1e3:.2f} ms, idle {sum(g for g in gaps if g > 0) / 1e3:.2f} ms")
d = collections.Counter(); t = collections.Counter()
for r in seg:
    du = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = "<6us" if du < 6 else "<12us" if du < 12 else "<50us" if du < 50 else ">=50us"
    d[k] += 1; t[k] += du
for k in ("<6us", "<12us", "<50us", ">=50us"): print(f"  {k:7s} {d[k]:5d} kernels {t[k] / 1e3:7.2f} ms")
big = sorted(((g, short(seg[i]["Kernel_Name"]), short(seg[i + 1]["Kernel_Name"])) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, x, y in big: print(f"  gap {g:7.1f} us  {x} -> {y}")
h = collections.Counter(); ht = collections.Counter()
for g in gaps:
    k = "<=0.5" if g <= 0.5 else "<2" if g < 2 else "<4" if g < 4 else "<8" if g < 8 else "<12" if g < 12 else ">=12"
    h[k] += 1; ht[k] += max(g, 0)
print("gaps (us):", {k: (h[k], round(ht[k] / 1e3, 2)) for k in ("<=0.5", "<2", "<4", "<8", "<12", ">=12")})
after = collections.Counter(); aft = collections.Counter()
for i, g in enumerate(gaps):
    if g >= 8: after[short(seg[i]["Kernel_Name"]) + " -> " + short(seg[i + 1]["Kernel_Name"])] += 1; aft[short(seg[i]["Kernel_Name"]) + " -> " + short(seg[i + 1]["Kernel_Name"])] += g
print("gaps >= 8 us by transition:")
for k, v in aft.most_common(20): print(f"  x{after[k]:3d} {v:7.1f} us  {k}")
c = collections.Counter(); ct = collections.Counter()
for r in seg:
    du = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if du < 12: c[short(r["Kernel_Name"])] += 1; ct[short(r["Kernel_Name"])] += du
print("short kernels (< 12 us) by name:")
for k, v in ct.most_common(25): print(f"  {k:42s} x{c[k]:4d} {v:8.1f} us")
PY
rm -rf gpurun_out/tr_gap
