#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p22"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 900 python tools/check_gemv_mx.py time > "$O/mx.log" 2>&1; echo "rc=$?"; grep -v amdgpu.ids "$O/mx.log" | tail -60
