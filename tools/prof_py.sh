#!/bin/bash
# rocprofv3 kernel summary of any python script.  usage: tools/prof_py.sh <script.py> [rows] [args...]
export TMPDIR=/tmp; R=$(pwd); S=$1; N=${2:-16}; shift; shift; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_py -o p -- python $R/$S "$@" > $R/gpurun_out/prof_py.log 2>&1
cd $R; tail -12 gpurun_out/prof_py.log; python tools/stats_table.py $(find gpurun_out/prof_py -name "*kernel_stats.csv") $N; rm -rf gpurun_out/prof_py
