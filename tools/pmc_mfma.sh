#!/bin/bash
# MFMA-utilisation PMC pass of bench.py for the dominant kernel -> gpurun_out/<tag>_mfma_util_pmc.json
TAG=${1:-r02e}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmcM_$TAG -o m -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2> $OUT/${TAG}_pmcM.err
cd $R
M=$(find $OUT/pmcM_$TAG -name "*counter_collection.csv" | head -1)
head -1 $M
python tools/pmc_mfma_summary.py $M $OUT/${TAG}_mfma_util_pmc.json
rm -rf $OUT/pmcM_$TAG
