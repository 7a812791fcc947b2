#!/bin/bash
# A/B of two builds of the library on ONE box: buddy_amd/libbuddy_hip_A.so and _B.so are copied over libbuddy_hip.so in turn (A B A B).
# usage: bash tools/ab_lib.sh [bench flags]
cd $(dirname $0)/..
for v in A B A B; do
  cp buddy_amd/libbuddy_hip_$v.so buddy_amd/libbuddy_hip.so
  python bench.py --steps 10 --warmup 3 --legs none --no-cpu-baseline --also-concurrent 0 --no-rccl-selftest "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
tp=j['conv3x3']['transform_passes']
print('$v', 'ms/step %.2f' % j['ms_per_step'], 'gemm us %.1f' % (j['roofline']['avg_launch_ms']*1e3), 'in GB/s %.0f out GB/s %.0f' % (tp['input_GBps'], tp['output_GBps']), 'gn GB/s %.0f share %.3f' % (j['roofline_hbm']['achieved'], j['roofline_hbm']['kernel_time_share_of_step']), 'op ms %.2f' % j['operator_update']['ms_per_step'], 'mfma box %.0f' % j['peaks']['measured_on_this_box']['bf16_mfma_tflops'])
"
done
cp buddy_amd/libbuddy_hip_B.so buddy_amd/libbuddy_hip.so
