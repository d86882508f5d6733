#!/bin/bash
# round-2 probe 10: GEMV launch shapes (diag lib) on C2 / C1 / the 14336^2 shape
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p10"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so
timeout -k 10 400 python tools/ab_kernels.py gemv 40 41 42 43 44 45 46 47 48 49 > "$O/ab_gemv.log" 2>&1; grep -v amdgpu.ids "$O/ab_gemv.log"
timeout -k 10 400 python tools/ab_kernels.py gemv_sq 40 41 42 43 47 49 > "$O/ab_gemv_sq.log" 2>&1; grep -v amdgpu.ids "$O/ab_gemv_sq.log"
