# usage: prof_one.sh <tag> ; env passes through
TAG=$1; R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --also-concurrent 0 > $R/gpurun_out/$TAG.json 2> $R/gpurun_out/$TAG.err
find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${TAG}_kernel_stats.csv \; ; rm -rf $R/gpurun_out/prof_$TAG; cd $R
python tools/stats_table.py gpurun_out/${TAG}_kernel_stats.csv | head -${2:-12}
