#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p5"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_stamp.so
timeout -k 10 200 python tools/stamp_gemm.py gemm 5 > "$O/stamp_c3.log" 2>&1; grep -v amdgpu.ids "$O/stamp_c3.log" | tail -8
timeout -k 10 200 python tools/stamp_gemm.py flux 4 > "$O/stamp_flux.log" 2>&1; grep -v amdgpu.ids "$O/stamp_flux.log" | tail -8
