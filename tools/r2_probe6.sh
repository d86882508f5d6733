#!/bin/bash
# round-2 probe 6: reads-first loop order + one-barrier NaN verdict: parity, then A/B on FLUX / shard / C3
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p6"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
for k in 30 31 32 33 4 5 2 15 17; do
  timeout -k 5 120 python tools/check_kernel.py $k >> "$O/check.log" 2>&1 || { echo "check_kernel $k FAILED rc=$?" | tee -a "$O/check.log"; tail -5 "$O/check.log"; exit 1; }
done
grep -v amdgpu.ids "$O/check.log"
timeout -k 10 300 python tools/ab_kernels.py flux 4 30 33 11 > "$O/ab_flux.log" 2>&1; grep -v amdgpu.ids "$O/ab_flux.log"
timeout -k 10 300 python tools/ab_kernels.py gemm 5 32 15 > "$O/ab_c3.log" 2>&1; grep -v amdgpu.ids "$O/ab_c3.log"
for k in 2 31 17; do timeout -k 10 120 python tools/time_shape.py 4096 3072 1536 $k bf16 >> "$O/shard.log" 2>&1; done
grep -v amdgpu.ids "$O/shard.log"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$O/pytest.log"
