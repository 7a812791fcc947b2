#!/bin/bash
# on the GPU box: correctness of the 16-bit attention kernels against the fp64 host reference, then timings at the long-form shape (whole launchers by
# HIP events; per kernel from a rocprofv3 kernel trace of the same run), per build variant
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$(pwd)
{
for P in tools/probes/attn16_probe_*; do
  echo "== $P"
  timeout 300 $P check 2 300 256 2
  timeout 300 $P check 1 131 64 1
  timeout 300 $P time 4 15008 256 2 5
  timeout 300 $P time 8 2048 256 2 10
  (cd /tmp && rm -rf pa16 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa16 -o t -- $R/$P time 4 15008 256 2 5 > /dev/null 2>&1; python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pa16/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "fa16" in n or "cvt16" in n or "stats16" in n:
        k = n[n.find("fa16"):][:16] if "fa16" in n else ("cvt16" if "cvt16" in n else "stats16")
        print(f"   {k:18s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:9.1f} us")
PY
)
done
} 2>&1 | tee gpurun_out/attn16_probe.log
