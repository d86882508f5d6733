#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p13"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so
timeout -k 10 300 python tools/check_gemv.py 58 59 60 62 63 64 > "$O/check.log" 2>&1 || { echo "check failed"; tail -5 "$O/check.log"; exit 1; }
grep -v amdgpu.ids "$O/check.log"
timeout -k 10 400 python tools/ab_kernels.py gemv 40 41 43 45 46 54 55 63 > "$O/ab_gemv.log" 2>&1; grep -v amdgpu.ids "$O/ab_gemv.log"
timeout -k 10 400 python tools/ab_kernels.py gemv_sq 41 46 52 54 55 63 > "$O/ab_gemv_sq.log" 2>&1; grep -v amdgpu.ids "$O/ab_gemv_sq.log"
for k in 1 61 58 59 60 62 64; do timeout -k 10 120 python tools/time_shape.py 1 4096 4096 $k >> "$O/c1.log" 2>&1; done; grep -v amdgpu.ids "$O/c1.log"
for k in 1 56 62 54; do timeout -k 10 120 python tools/time_shape.py 1 8192 8192 $k >> "$O/c1.log" 2>&1; done; grep -v amdgpu.ids "$O/c1.log" | tail -4
for k in 1 58 59 64; do timeout -k 10 120 python tools/time_shape.py 1 4096 14336 $k >> "$O/c1.log" 2>&1; done; grep -v amdgpu.ids "$O/c1.log" | tail -4
