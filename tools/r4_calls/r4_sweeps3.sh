#!/bin/bash
set -uo pipefail
O=gpurun_out/r4t; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
MS=200,224,256,288,320,384,448,512 timeout -k 10 700 python tools/sweep_regret.py 71 130 > $O/regret71_midm.txt 2>&1; echo "midm 71 rc=$? $(tail -1 $O/regret71_midm.txt)"
MS=200,256,300,320,384,400,512 DIMS=ext timeout -k 10 700 python tools/sweep_regret.py 72 110 > $O/regret72_midm_ext.txt 2>&1; echo "midm ext rc=$? $(tail -1 $O/regret72_midm_ext.txt)"
MS=224,256,320,384,512 NOWS=1 timeout -k 10 600 python tools/sweep_regret.py 73 70 > $O/regret73_midm_nows.txt 2>&1; echo "midm nows rc=$? $(tail -1 $O/regret73_midm_nows.txt)"
for f in $O/regret7*.txt; do echo "== $f"; grep "^#  " $f | head -6; done
