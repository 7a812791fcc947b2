#!/bin/bash
# same-box A/B of bench legs under an environment switch.  usage: tools/ab_legs.sh VAR v1 v2 [legs] [reps]
VAR=$1; A=$2; B=$3; LEGS=${4:-blind_b1,informed_b1}; REPS=${5:-2}
for i in $(seq $REPS); do for v in $A $B; do
  env $VAR=$v python bench.py --no-cpu-baseline --also-concurrent 0 --legs $LEGS --no-rccl-selftest --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', round(d['ms_per_step'], 2), {k: round(x['ms_per_step'], 2) for k, x in d['legs'].items()})"
done; done
