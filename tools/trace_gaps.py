"""Idle time between consecutive kernels from a rocprofv3 --kernel-trace CSV: per kernel name, total duration and total gap BEFORE it.
usage: python tools/trace_gaps.py <kernel_trace.csv> [top]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(lambda: [0, 0, 0])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("buddy::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    r["_n"] = n
    a = agg[n]; a[0] += 1; a[1] += e - s
    if prev_end is not None:
        g = s - prev_end
        if 0 < g < 200000:          # ignore host-side pauses (> 0.2 ms)
            a[2] += g
    prev_end = max(prev_end or 0, e)
tot_d = sum(a[1] for a in agg.values()); tot_g = sum(a[2] for a in agg.values())
print(f"kernels {len(rows)}  busy {tot_d/1e6:.2f} ms  short gaps {tot_g/1e6:.2f} ms")
print(f"{'kernel':48s} {'calls':>6s} {'dur_ms':>9s} {'gap_ms':>9s} {'avg_dur_us':>10s} {'avg_gap_us':>10s}")
for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:top]:
    print(f"{n:48s} {c:6d} {d/1e6:9.3f} {g/1e6:9.3f} {d/c/1e3:10.1f} {g/c/1e3:10.1f}")

# where do the blit copies come from: (previous kernel -> copy) pairs and copy sizes
pairs = collections.Counter(); sizes = collections.Counter()
for a, b in zip(rows, rows[1:]):
    if "copyBuffer" in b["Kernel_Name"]:
        pairs[a["_n"]] += 1; sizes[b["Grid_Size_X"]] += 1
print("\ncopyBuffer predecessors:", pairs.most_common(12))
print("copyBuffer grid sizes:", sizes.most_common(8))
