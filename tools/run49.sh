#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p49"; mkdir -p "$O"; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.log"
bash tools/sweep_large.sh > "$O/large.txt" 2>&1; grep "kernel=0 " "$O/large.txt"
for shape in "2048 4096 4096" "1024 8192 8192" "4096 3072 1536" "1536 3072 4096" "1024 4096 4096"; do timeout -k 10 120 python tools/time_shape.py $shape 0 bf16 40 2>&1 | grep -v amdgpu.ids | tail -1; done
