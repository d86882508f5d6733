#!/bin/bash
set -uo pipefail
O=gpurun_out/r4s; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 600 python -m pytest tests/test_gpu_patch.py -m gpu -x -q > $O/pytest_patch.log 2>&1; echo "pytest patch rc=$?"; tail -3 $O/pytest_patch.log
for seed in 57 58 59; do DIMS=ext timeout -k 10 600 python tools/sweep_regret.py $seed 140 > $O/regret${seed}_ext.txt 2>&1; echo "ext $seed rc=$? $(tail -1 $O/regret${seed}_ext.txt)"; done
OUT=f32 DIMS=ext timeout -k 10 600 python tools/sweep_regret.py 60 100 > $O/regret60_ext_f32.txt 2>&1; echo "ext f32 rc=$? $(tail -1 $O/regret60_ext_f32.txt)"
NOWS=1 DIMS=ext timeout -k 10 600 python tools/sweep_regret.py 64 100 > $O/regret64_ext_nows.txt 2>&1; echo "nows ext rc=$? $(tail -1 $O/regret64_ext_nows.txt)"
for f in $O/regret5[789]_ext.txt $O/regret64_ext_nows.txt; do echo "== $f"; grep "^#  " $f | head -8; done
