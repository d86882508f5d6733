#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p28"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so
for k in 20 91 94 96; do timeout -k 10 300 python tools/check_gemm256.py $k > "$O/check$k.log" 2>&1; rc=$?; echo "check $k rc=$rc"; grep -v amdgpu.ids "$O/check$k.log" | tail -2; [ $rc -eq 0 ] || exit 1; done
timeout -k 10 600 python tools/ab_kernels.py flux 4 20 91 92 93 94 95 96 > "$O/ab.log" 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids "$O/ab.log" | tail -9
