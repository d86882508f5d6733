#!/bin/bash
set -uo pipefail
O=gpurun_out/r4d; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for seed in 31 32; do
  timeout -k 10 900 python tools/sweep_regret.py $seed 140 > $O/regret$seed.txt 2>&1; echo "regret $seed rc=$?"; tail -1 $O/regret$seed.txt
done
OUT=f32 timeout -k 10 600 python tools/sweep_regret.py 33 80 > $O/regret33_f32.txt 2>&1; echo "regret f32 rc=$?"; tail -1 $O/regret33_f32.txt
NOWS=1 timeout -k 10 600 python tools/sweep_regret.py 34 80 > $O/regret34_nows.txt 2>&1; echo "regret nows rc=$?"; tail -1 $O/regret34_nows.txt
for f in $O/regret3*.txt; do echo "== $f"; grep "^#  " $f | head -8; done
