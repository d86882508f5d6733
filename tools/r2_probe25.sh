#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p25"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so timeout -k 10 600 python tools/ab_kernels.py flux 4 20 81 82 83 84 85 86 87 88 89 90 > "$O/ab.log" 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids "$O/ab.log" | tail -14
