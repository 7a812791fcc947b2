#!/bin/bash
# sub-batches S in {1, 2, 4} x batch B in {8, 16} of the blind step (utterances of a batch sampled as S concurrent sub-batches on S
# HIP streams, replicas of one network).  One line per cell: ms/step, utterance-steps/s.   usage: bash tools/sweep_sub_batches.sh > profiles/rNN_sub_batch_sweep.txt
FLAGS="--steps 8 --warmup 3 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest"
for B in 8 16; do
  for S in 1 2 4; do
    python bench.py $FLAGS --batch $B --sub-batches $S 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B=$B S=$S  ms_per_step %.2f  utterance-steps/s %.1f  ms per utterance-step %.3f' % (d['ms_per_step'], d['value'], d['ms_per_step'] / $B))"
  done
done
