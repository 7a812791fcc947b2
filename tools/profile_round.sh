#!/bin/bash
# One gpurun call: default bench line (+ cpu baseline), rocprofv3 kernel stats of the same command, PMC traffic passes, long-form line.
# usage: bash tools/profile_round.sh <tag>      (outputs under gpurun_out/, copy what is to be judged into profiles/)
TAG=${1:-r02a}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_bench_N1.json 2> $OUT/${TAG}_bench_N1.err; tail -c 600 $OUT/${TAG}_bench_N1.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_prof.err
find $OUT/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_bench_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcF_$TAG -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2> $OUT/${TAG}_pmcF.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcW_$TAG -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > /dev/null 2> $OUT/${TAG}_pmcW.err
cd $R
F=$(find $OUT/pmcF_$TAG -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmcW_$TAG -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F $W $OUT/${TAG}_conv_traffic_pmc.json | head -c 400; echo
rm -rf $OUT/pmcF_$TAG $OUT/pmcW_$TAG $OUT/prof_$TAG
python bench.py --length 480000 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > $OUT/${TAG}_bench_longform_480000_B4.json 2> $OUT/${TAG}_longform.err; tail -c 300 $OUT/${TAG}_bench_longform_480000_B4.json; echo
python bench.py --gemm fp32 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest > $OUT/${TAG}_bench_N1_gemm_fp32.json 2> /dev/null; tail -c 200 $OUT/${TAG}_bench_N1_gemm_fp32.json; echo
bash tools/pmc_mfma.sh $TAG > /dev/null 2>&1
python tools/stats_table.py $OUT/${TAG}_bench_kernel_stats.csv | head -40
