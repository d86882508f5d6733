#!/bin/bash
# round-2 probe 7: staggered wave groups (MODE 2): parity + A/B on FLUX and its 128x128 shard kernel
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p7"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
for k in 34 35 36 37; do
  timeout -k 5 120 python tools/check_kernel.py $k >> "$O/check.log" 2>&1 || { echo "check_kernel $k FAILED rc=$?" | tee -a "$O/check.log"; tail -5 "$O/check.log"; exit 1; }
done
grep -v amdgpu.ids "$O/check.log"
timeout -k 10 300 python tools/ab_kernels.py flux 4 33 34 35 > "$O/ab_flux.log" 2>&1; grep -v amdgpu.ids "$O/ab_flux.log"
for k in 2 36 37; do timeout -k 10 120 python tools/time_shape.py 4096 3072 1536 $k bf16 >> "$O/shard.log" 2>&1; done
for k in 4 34 2 36; do timeout -k 10 120 python tools/time_shape.py 8192 8192 8192 $k bf16 8 >> "$O/shard.log" 2>&1; done
grep -v amdgpu.ids "$O/shard.log"
