#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p27"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 300 python tools/check_gemm256.py > "$O/check.log" 2>&1; rc=$?; echo "check rc=$rc"; grep -v amdgpu.ids "$O/check.log" | tail -4
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/ab_kernels.py flux 4 20 > "$O/ab.log" 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids "$O/ab.log" | tail -3
FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_stamp.so timeout -k 10 300 python tools/stamp_gemm256.py flux 20 > "$O/stamp.log" 2>&1; echo "rc=$?"; grep -v amdgpu.ids "$O/stamp.log" | tail -12
