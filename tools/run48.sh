#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p48"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 300 python tools/check_gemm256.py 21 > "$O/check21.log" 2>&1; rc=$?; echo "check 21 rc=$rc"; grep -v amdgpu.ids "$O/check21.log" | tail -4
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/check_gemm256.py > "$O/check20.log" 2>&1; echo "check 20 rc=$?"; grep -v amdgpu.ids "$O/check20.log" | tail -1
for shape in "2048 4096 4096" "1024 4096 4096" "4096 3072 1536" "1536 3072 4096" "2048 4096 14336" "4096 4096 4096" "4096 3072 12288" "1024 8192 8192" "3072 3072 3072"; do for kid in 0 2 20 21; do timeout -k 10 120 python tools/time_shape.py $shape $kid bf16 40 2>&1 | grep -v amdgpu.ids | tail -1; done; done
