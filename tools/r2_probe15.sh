#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p15"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_diag.so
for k in 25 26 27 28; do timeout -k 5 120 python tools/check_kernel.py $k >> "$O/check.log" 2>&1 || { echo "check $k FAILED"; tail -5 "$O/check.log"; exit 1; }; done
grep -v amdgpu.ids "$O/check.log"
timeout -k 10 400 python tools/ab_kernels.py gemm 5 15 25 26 28 203 208 > "$O/ab.log" 2>&1; grep -v amdgpu.ids "$O/ab.log"
for k in 2 17 27; do timeout -k 10 120 python tools/time_shape.py 4096 3072 1536 $k bf16 >> "$O/shard.log" 2>&1; done; grep -v amdgpu.ids "$O/shard.log"
