"""Idle time between consecutive kernels of a rocprofv3 kernel trace (one stream of work): where the GPU waits for the host.
usage: python tools/gap_report.py <*_kernel_trace.csv> [skip_fraction]   (the first skip_fraction of the trace = start-up, default 0.5)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
marker = sys.argv[3] if len(sys.argv) > 3 else None
if marker:                                   # window = from the (skip)-th to the last launch of a once-per-step kernel: whole sampler steps
    m = [e[0] for e in ev if marker in e[2]]
    k0 = int(skip)
    ev = [e for e in ev if m[k0] <= e[0] < m[-1]]
    print(f"marker '{marker}': {len(m)} launches, window = steps {k0}..{len(m) - 1} -> {len(m) - 1 - k0} steps")
    nsteps = len(m) - 1 - k0
else:
    t0, t1 = ev[0][0], ev[-1][1]
    cut = t0 + skip * (t1 - t0)
    ev = [e for e in ev if e[0] >= cut]
    nsteps = 1
busy = sum(e[1] - e[0] for e in ev)
span = ev[-1][1] - ev[0][0]
gaps = []
end = ev[0][1]
for s, e, n in ev[1:]:
    if s > end:
        gaps.append((s - end, n))
    end = max(end, e)
tot = sum(g for g, _ in gaps)
print(f"per step: span {span/1e6/nsteps:.2f} ms, busy {busy/1e6/nsteps:.2f} ms, kernels {len(ev)/nsteps:.0f}")
print(f"kernels {len(ev)}  span {span/1e6:.1f} ms  busy(sum) {busy/1e6:.1f} ms  idle {tot/1e6:.2f} ms = {100*tot/span:.2f} % of the span")
for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
    sel = [g for g, _ in gaps if lo <= g < hi]
    print(f"  gaps {lo/1e3:>6.0f}-{hi/1e3:<8.0f} us: {len(sel):6d}  total {sum(sel)/1e6:7.2f} ms")
big = {}
for g, n in gaps:
    if g >= 2e4:
        k = n.split("(")[0][-60:]
        big[k] = big.get(k, [0, 0]); big[k][0] += 1; big[k][1] += g
for k, (c, g) in sorted(big.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  before {k:60s} {c:5d} gaps  {g/1e6:7.2f} ms")

if marker:                                   # launches per step by kernel, largest time first
    agg = {}
    for st, en, n in ev:
        k = n.split("(")[0][-70:]
        a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += en - st
    print("per step (launches, ms):")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 0]:
        print(f"  {k:70s} {c / nsteps:8.1f} {t / 1e6 / nsteps:8.3f}")
