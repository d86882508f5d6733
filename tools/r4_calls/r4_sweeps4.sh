#!/bin/bash
set -uo pipefail
O=gpurun_out/r4t; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
MS=288,384,448,512,640,768,896,1024 timeout -k 10 900 python tools/sweep_regret.py 74 130 > $O/regret74_midm.txt 2>&1; echo "midm 74 rc=$? $(tail -1 $O/regret74_midm.txt)"
MS=400,512,600,768,1000,1024 DIMS=ext timeout -k 10 900 python tools/sweep_regret.py 75 110 > $O/regret75_midm_ext.txt 2>&1; echo "midm ext 75 rc=$? $(tail -1 $O/regret75_midm_ext.txt)"
MS=200,256,320,384,512,768,1024 NOWS=1 DIMS=ext timeout -k 10 700 python tools/sweep_regret.py 76 80 > $O/regret76_midm_ext_nows.txt 2>&1; echo "midm ext nows rc=$? $(tail -1 $O/regret76_midm_ext_nows.txt)"
for f in $O/regret74_midm.txt $O/regret75_midm_ext.txt; do echo "== $f"; grep "^#  " $f | head -8; done
