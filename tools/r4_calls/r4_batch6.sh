#!/bin/bash
set -uo pipefail
O=gpurun_out/r4f; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
export FP8MI_LIB_PATH=$PWD/fp8-mps-metal_amd/libfp8mi_diag.so
timeout -k 10 300 python tools/check_kernel.py 142 143 144 148 145 146 147 > $O/check.txt 2>&1; echo "check rc=$?"; cat $O/check.txt | tail -8
timeout -k 10 300 python tools/ab_kernels.py gemm 5 142 143 144 148 38 > $O/ab_gemm.txt 2>&1; echo "ab gemm rc=$?"; cat $O/ab_gemm.txt | grep kernel
timeout -k 10 300 python tools/ab_kernels.py wide 25 134 145 146 147 > $O/ab_wide.txt 2>&1; echo "ab wide rc=$?"; cat $O/ab_wide.txt | grep kernel
