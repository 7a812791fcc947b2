#!/bin/bash
# wave-cycle / MFMA-busy counters of the bf16x3 GEMM alone on one shape -> stdout.  usage: tools/pmc_wgemm.sh Mt N K [P]
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/pmcW
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmcW -o m -- python $R/tools/wgemm_one.py $1 $2 $3 ${4:-64} 3 > /dev/null 2> $OUT/pmcW.err
cd $R
M=$(find $OUT/pmcW -name "*counter_collection.csv" | head -1)
python tools/pmc_mfma_summary.py $M $OUT/pmcW_$1_$2_$3.json wgemm_bf16x3_kernel
rm -rf $OUT/pmcW
