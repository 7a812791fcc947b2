#!/bin/bash
# rocprofv3 kernel stats of the long-form configuration (BASELINE configs[4] one-GPU slice: B = 4 x 480 000 samples), fp32 attention (auto -> online-softmax kernels).
# usage: bash tools/prof_longform.sh <tag> [attention]      (attention = auto | f16 | bf16 ...; with one, the outputs are named <tag>_longform_<attention>_*)
TAG=${1:-r04e}
ATT=${2:-}
if [ -n "$ATT" ]; then ATTARG="--attention $ATT"; TAG=${TAG}_longform_$ATT; SUF=""; else ATTARG=""; SUF="_longform"; fi
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lf_$TAG -o lf -- python $R/bench.py --length 480000 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest $ATTARG > $OUT/${TAG}${SUF}_under_rocprof.json 2> $OUT/${TAG}_lf.err
cd $R
find $OUT/prof_lf_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}${SUF}_kernel_stats.csv \;
rm -rf $OUT/prof_lf_$TAG
python tools/stats_table.py $OUT/${TAG}${SUF}_kernel_stats.csv | head -16
