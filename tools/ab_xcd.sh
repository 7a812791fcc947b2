set -x
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "winograd or conv3 or gemm" 2>&1 | tail -3
for cfg in "A=1" "BUDDY_W6_XCD=0" "BUDDY_WGEMM_XCDPOS=0" "A=2"; do
  env $cfg python bench.py --steps 10 --warmup 3 --legs none --no-cpu-baseline --also-concurrent 0 --no-rccl-selftest 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
tp=j['conv3x3']['transform_passes']
print('$cfg', 'ms/step %.2f' % j['ms_per_step'], 'gemm us %.1f' % (j['roofline']['avg_launch_ms']*1e3), 'in GB/s %.0f out GB/s %.0f' % (tp['input_GBps'], tp['output_GBps']), 'mfma box %.0f' % j['peaks']['measured_on_this_box']['bf16_mfma_tflops'])
"
done
