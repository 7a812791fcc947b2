#!/bin/bash
# A/B of one environment switch on one box: alternating bench runs (main region only), ms/step of each.
# usage: bash tools/ab_env.sh VAR A_VALUE B_VALUE [rounds] [extra bench flags]
VAR=$1; A=$2; B=$3; R=${4:-2}; shift 4
FLAGS="--steps 10 --warmup 3 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest $@"
for i in $(seq $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py $FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', 'ms_per_step', round(d['ms_per_step'],2), 'wgemm_us', round(1e3*d['roofline'].get('ms_per_launch',0),1) if 'ms_per_launch' in d['roofline'] else '')"
  done
done
