#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p12"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so
timeout -k 10 300 python tools/check_gemv.py 40 50 51 52 53 54 55 56 57 > "$O/check.log" 2>&1 || { echo "check failed"; tail -5 "$O/check.log"; exit 1; }
grep -v amdgpu.ids "$O/check.log"
timeout -k 10 400 python tools/ab_kernels.py gemv 40 50 51 52 53 54 55 48 > "$O/ab_gemv.log" 2>&1; grep -v amdgpu.ids "$O/ab_gemv.log"
timeout -k 10 400 python tools/ab_kernels.py gemv_sq 40 50 51 52 53 > "$O/ab_gemv_sq.log" 2>&1; grep -v amdgpu.ids "$O/ab_gemv_sq.log"
for k in 1 56 57 51 55 52; do timeout -k 10 120 python tools/time_shape.py 1 4096 4096 $k >> "$O/c1.log" 2>&1; done; grep -v amdgpu.ids "$O/c1.log"
