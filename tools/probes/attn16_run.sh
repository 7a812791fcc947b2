#!/bin/bash
# on the GPU box: correctness of the 16-bit attention kernels against the fp64 host reference, then timings at the long-form shape
P=tools/probes/attn16_probe
mkdir -p gpurun_out
{
for prec in 1 2; do
  timeout 300 $P check 2 300 256 $prec
  timeout 300 $P check 1 131 64 $prec
  timeout 300 $P check 1 320 128 $prec
done
timeout 300 $P time 4 15008 256 2 5
timeout 300 $P time 4 15008 256 1 5
timeout 300 $P time 8 2048 256 2 10
} 2>&1 | tee gpurun_out/attn16_probe.log
