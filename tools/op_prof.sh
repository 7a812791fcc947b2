#!/bin/bash
# rocprofv3 kernel summary of the operator update alone (tools/op_only.py).  usage: tools/op_prof.sh [rows]
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_op -o op -- python $R/tools/op_only.py 8 10 > /dev/null 2>&1
cd $R; python tools/stats_table.py $(find gpurun_out/prof_op -name "*kernel_stats.csv") ${1:-14}; rm -rf gpurun_out/prof_op
