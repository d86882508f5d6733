#!/bin/bash
set -uo pipefail
O=gpurun_out/r4g; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=$PWD/fp8-mps-metal_amd/libfp8mi_diag.so
timeout -k 10 300 python tools/check_kernel.py 142 144 > $O/check.txt 2>&1; echo "check rc=$?"; tail -3 $O/check.txt
timeout -k 10 400 python tools/ab_kernels.py gemm 5 39 142 143 144 145 38 > $O/ab_gemm.txt 2>&1; echo "ab gemm rc=$?"; grep kernel $O/ab_gemm.txt
