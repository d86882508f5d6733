#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p11"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so
timeout -k 10 400 python tools/ab_kernels.py gemv 40 49 43 48 > "$O/ab_gemv.log" 2>&1; grep -v amdgpu.ids "$O/ab_gemv.log"
