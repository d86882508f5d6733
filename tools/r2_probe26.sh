#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p26"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_stamp.so timeout -k 10 300 python tools/stamp_gemm256.py flux 20 > "$O/stamp.log" 2>&1; echo "rc=$?"; grep -v amdgpu.ids "$O/stamp.log" | tail -14
