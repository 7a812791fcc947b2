#!/bin/bash
# A/B of the Winograd-domain GEMM arithmetics (bf16x3 | f16x2 | fp32): unit test, fixtures, precision budget vs float64, bench line.
# usage: bash tools/eval_gemm_modes.sh <tag>   (outputs under gpurun_out/)
TAG=${1:-r05g}
OUT=gpurun_out; mkdir -p $OUT
python -m pytest tests/test_hip_kernels.py -q -x -s -k "winograd_domain_gemm" 2>&1 | grep -v "^$" | tail -15 > $OUT/${TAG}_unit.txt; tail -12 $OUT/${TAG}_unit.txt
for g in bf16x3 f16x2; do
  BUDDY_GEMM=$g python -m pytest tests/test_hip_fullsize.py -q -x -s -k "precision_budget_one" 2>&1 | grep "sigma\|passed\|failed\|Error" > $OUT/${TAG}_budget_$g.txt; cat $OUT/${TAG}_budget_$g.txt
  BUDDY_GEMM=$g python -m pytest tests/test_hip_network.py -q -x -s -k "vs_golden" 2>&1 | grep "forward\|passed\|failed\|Error" | tail -12 > $OUT/${TAG}_golden_$g.txt; tail -6 $OUT/${TAG}_golden_$g.txt
  python bench.py --gemm $g --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > $OUT/${TAG}_bench_$g.json 2> $OUT/${TAG}_bench_$g.err
  python - <<EOF
import json
d=json.loads([l for l in open("$OUT/${TAG}_bench_$g.json") if l.startswith("{")][-1])
print("$g", d["value"], d["ms_per_step"], d["roofline"].get("avg_launch_ms"), d["roofline"].get("share_of_step"))
EOF
done
