#!/bin/bash
# Build libbuddy_hip.so for gfx950 in-tree (buddy_amd/libbuddy_hip.so).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libbuddy_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result"
mkdir -p obj
pids=()
for f in options igemm wgemm wprep wino wino4 wino6 ops attn attn16 sampler net operator wpe capi; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || [ net.h -nt obj/$f.o ] || [ ../../include/buddy_hip.h -nt obj/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT obj/options.o obj/igemm.o obj/wgemm.o obj/wprep.o obj/wino.o obj/wino4.o obj/wino6.o obj/ops.o obj/attn.o obj/attn16.o obj/sampler.o obj/net.o obj/operator.o obj/wpe.o obj/capi.o
echo "built $(readlink -f $OUT)"
