#!/bin/bash
# round-2 probe 3: what bounds the C3 K loop - ablations of the producer/consumer kernel, padded strides, warm weights
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p3"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so timeout -k 10 400 python tools/ab_kernels.py gemm 15 201 202 203 204 205 206 207 > "$O/abl.log" 2>&1; grep -v amdgpu.ids "$O/abl.log"
for pad in 0 64 128 256 512; do PAD=$pad timeout -k 10 120 python tools/time_shape.py 512 4096 4096 15 >> "$O/pad.log" 2>&1; done
for pad in 0 256; do PAD=$pad timeout -k 10 120 python tools/time_shape.py 512 4096 4096 5 >> "$O/pad.log" 2>&1; done
NB=1 timeout -k 10 120 python tools/time_shape.py 512 4096 4096 15 >> "$O/pad.log" 2>&1
NB=1 PAD=256 timeout -k 10 120 python tools/time_shape.py 512 4096 4096 15 >> "$O/pad.log" 2>&1
grep -v amdgpu.ids "$O/pad.log"
