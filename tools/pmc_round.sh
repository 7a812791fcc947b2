#!/bin/bash
# PMC traffic passes of bench.py (separate FETCH_SIZE / WRITE_SIZE runs, --kernel-trace only) -> gpurun_out/<tag>_conv_traffic_pmc.json
TAG=${1:-r02a}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcF_$TAG -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2> $OUT/${TAG}_pmcF.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcW_$TAG -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2> $OUT/${TAG}_pmcW.err
cd $R
F=$(find $OUT/pmcF_$TAG -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmcW_$TAG -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W $OUT/${TAG}_conv_traffic_pmc.json | head -c 600; echo
rm -rf $OUT/pmcF_$TAG $OUT/pmcW_$TAG
