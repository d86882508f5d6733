#!/bin/bash
# staggered (product) vs lockstep 256x256 loop in two otherwise identical PRODUCT builds, alternating, same box
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p18"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
for rep in 1 2 3; do
  for lib in libfp8mi.so libfp8mi_lock.so; do
    FP8MI_LIB_PATH=fp8-mps-metal_amd/$lib timeout -k 10 200 python tools/ab_kernels.py flux 4 2>/dev/null | grep -v amdgpu | sed "s/^/$lib /" >> "$O/ab.log"
  done
done
for lib in libfp8mi.so libfp8mi_lock.so; do
  FP8MI_LIB_PATH=fp8-mps-metal_amd/$lib timeout -k 10 200 python tools/time_shape.py 8192 8192 8192 4 bf16 8 2>/dev/null | grep -v amdgpu | sed "s/^/$lib /" >> "$O/ab.log"
  FP8MI_LIB_PATH=fp8-mps-metal_amd/$lib timeout -k 10 200 python tools/time_shape.py 4096 4096 4096 4 bf16 16 2>/dev/null | grep -v amdgpu | sed "s/^/$lib /" >> "$O/ab.log"
done
cat "$O/ab.log"
