#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p8"; mkdir -p "$O"; cd "$R"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -15 "$O/pytest.log"
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 300 python tools/ab_kernels.py gemm 5 0 > "$O/ab.log" 2>&1
timeout -k 10 300 python tools/ab_kernels.py flux 4 0 >> "$O/ab.log" 2>&1
for n in 4096 8192 16384; do timeout -k 10 120 python tools/time_shape.py 1 14336 $n 0 >> "$O/gemv.log" 2>&1; done
timeout -k 10 120 python tools/time_shape.py 1 4096 4096 0 >> "$O/gemv.log" 2>&1
timeout -k 10 120 python tools/time_shape.py 1 14336 14336 0 >> "$O/gemv.log" 2>&1
grep -v amdgpu.ids "$O/ab.log" "$O/gemv.log"
