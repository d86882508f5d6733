#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p21"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 300 python tools/dbg_mx.py > "$O/dbg.log" 2>&1; echo "dbg rc=$?"; grep -v amdgpu.ids "$O/dbg.log" | tail -12
FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so timeout -k 10 700 python tools/check_gemv_mx.py time > "$O/mx.log" 2>&1; echo "rc=$?"; grep -v amdgpu.ids "$O/mx.log" | tail -20
