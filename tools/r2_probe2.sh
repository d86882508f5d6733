#!/bin/bash
# round-2 probe 2: producer/consumer GEMM kernels - parity, then A/B timing against the ring kernels
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p2"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
for k in 15 16 17 20 21 22 24 5 2; do
  timeout -k 5 90 python tools/check_kernel.py $k >> "$O/check.log" 2>&1 || { echo "check_kernel $k FAILED/timeout rc=$?" | tee -a "$O/check.log"; tail -5 "$O/check.log"; exit 1; }
done
cat "$O/check.log" | grep -v amdgpu.ids
timeout -k 10 300 python tools/ab_kernels.py gemm 5 15 20 22 24 17 21 > "$O/ab_c3.log" 2>&1; grep -v amdgpu.ids "$O/ab_c3.log"
for k in 17 21 2; do SPLIT=2 timeout -k 10 120 python tools/time_shape.py 512 4096 4096 $k >> "$O/split.log" 2>&1; done
for k in 5 15 20; do SPLIT=2 timeout -k 10 120 python tools/time_shape.py 512 4096 4096 $k >> "$O/split.log" 2>&1; done
grep -v amdgpu.ids "$O/split.log"
# the FLUX shard (4096 x 3072 x 1536, bf16) and the split-K decode shape
for k in 2 17 21; do timeout -k 10 120 python tools/time_shape.py 4096 3072 1536 $k bf16 >> "$O/shard.log" 2>&1; done
for k in 14 16 0; do timeout -k 10 120 python tools/time_shape.py 64 14336 4096 $k bf16 >> "$O/shard.log" 2>&1; done
grep -v amdgpu.ids "$O/shard.log"
