#!/bin/bash
# AUTO against the forced tile kernels on mid-size shapes (2 = 128x128 ring, 20 = 256x256 one wave per SIMD, 21 = its 256x128 form).
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HIP_FORCE_DEV_KERNARG=1
for shape in "2048 4096 4096" "1024 4096 4096" "1024 8192 8192" "4096 3072 1536" "1536 3072 4096" "3072 3072 3072" "16384 1024 8192" "2048 2048 8192" "1536 4096 6144" "512 4096 8192" "768 3072 3072" "8192 1024 2048" "2048 512 4096"; do
  for kid in 0 2 20 21; do
    timeout -k 10 120 python tools/time_shape.py $shape $kid bf16 30 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
