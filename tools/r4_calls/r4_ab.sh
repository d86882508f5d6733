#!/bin/bash
set -uo pipefail
O=gpurun_out/r4v; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 300 python tools/ab_kernels.py gemm 5 14 0 > $O/ab_c3.txt 2>&1; echo "rc=$?"; grep kernel $O/ab_c3.txt
timeout -k 10 900 python -m pytest tests/test_gpu_dispatch_regret.py -m gpu -q > $O/pytest_regret.log 2>&1; echo "regret rc=$?"; tail -4 $O/pytest_regret.log
