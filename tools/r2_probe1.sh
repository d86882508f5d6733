#!/bin/bash
# round-2 probe 1: GPU test suite, kernarg placement experiment, stamp shares of the C3 kernel
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p1"; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest.log"
tail -5 "$O/pytest.log"
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v" | tee -a "$O/kernarg.log"
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python tools/time_quantize.py 64 f32 >> "$O/kernarg.log" 2>&1
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python tools/time_shape.py 512 4096 4096 0 >> "$O/kernarg.log" 2>&1
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python tools/time_shape.py 1 14336 4096 0 >> "$O/kernarg.log" 2>&1
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python tools/time_shape.py 1 4096 4096 0 >> "$O/kernarg.log" 2>&1
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 120 python tools/time_shape.py 128 128 64 5 >> "$O/kernarg.log" 2>&1
done
cat "$O/kernarg.log"
FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_stamp.so timeout -k 10 200 python tools/stamp_gemm.py gemm 5 > "$O/stamp_c3.log" 2>&1; cat "$O/stamp_c3.log"
