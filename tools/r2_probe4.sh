#!/bin/bash
# round-2 probe 4: stamp shares of the ring kernels after kernarg pinning (C3 128x64, FLUX 256x256, shard 128x128)
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p4"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_stamp.so
timeout -k 10 200 python tools/stamp_gemm.py gemm 5 > "$O/stamp_c3.log" 2>&1; grep -v amdgpu.ids "$O/stamp_c3.log"
timeout -k 10 200 python tools/stamp_gemm.py flux 4 > "$O/stamp_flux.log" 2>&1; grep -v amdgpu.ids "$O/stamp_flux.log"
timeout -k 10 200 python tools/stamp_gemm.py flux 2 > "$O/stamp_flux128.log" 2>&1; grep -v amdgpu.ids "$O/stamp_flux128.log"
unset FP8MI_LIB_PATH
timeout -k 10 300 python tools/ab_kernels.py flux 4 2 17 21 > "$O/ab_flux.log" 2>&1; grep -v amdgpu.ids "$O/ab_flux.log"
