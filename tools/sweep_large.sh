#!/bin/bash
# Per-dispatch time of large scaled_mm shapes: AUTO against the forced tile kernels (4 = 256x256 ring, 2 = 128x128, 20 = 256x256
# one-wave-per-SIMD).  bash tools/sweep_large.sh > profiles/rNN_large_shapes.txt   (on the GPU box)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HIP_FORCE_DEV_KERNARG=1
for shape in "4096 3072 12288" "4096 12288 3072" "4096 3072 9216" "8192 8192 8192" "6144 6144 6144" "4096 4096 4096" "2048 4096 4096" "1024 4096 14336" "16384 1024 8192" "4096 3072 3072" "4608 3072 12288"; do
  for kid in 0 4 2 20; do
    timeout -k 10 120 python tools/time_shape.py $shape $kid bf16 30 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
