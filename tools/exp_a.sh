#!/bin/bash
# experiment batch A (one gpurun call): quick parity tests, cache probe, chunked-Winograd A/B, operator FFT A/B, short bench
mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py tests/test_hip_operator.py tests/test_hip_operator_golden.py tests/test_hip_network.py -m gpu -q -x > gpurun_out/expA_tests.log 2>&1; tail -15 gpurun_out/expA_tests.log
python tools/mall_probe.py > gpurun_out/expA_mall.log 2>&1; cat gpurun_out/expA_mall.log
for c in 0 1024 2048 4096; do echo "== BUDDY_W4_CHUNK=$c"; BUDDY_W4_CHUNK=$c python tools/wino4_one.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/expA_w4chunk.log 2>&1; cat gpurun_out/expA_w4chunk.log
for f in 1 0; do echo "== BUDDY_OP_FFT=$f"; BUDDY_OP_FFT=$f python tools/op_only.py 2>&1 | tail -3; done > gpurun_out/expA_opfft.log 2>&1; cat gpurun_out/expA_opfft.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/expA_bench.json 2> gpurun_out/expA_bench.err; cat gpurun_out/expA_bench.json
