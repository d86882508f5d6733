#!/bin/bash
# Live regret of the SHIPPED dispatch constants on seeds no fit or fixture has seen (cold weights, shuffled kernel order): standard shapes, mid-M, small dimensions, no workspace.
set -uo pipefail
O=gpurun_out/r4y; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for seed in 101 102; do timeout -k 10 600 python tools/sweep_regret.py $seed 140 > $O/std$seed.txt 2>&1; echo "std $seed rc=$?"; tail -1 $O/std$seed.txt; done
MS=200,256,320,384,448,512,640,768,896,1024 timeout -k 10 600 python tools/sweep_regret.py 103 140 > $O/mid103.txt 2>&1; echo "mid rc=$?"; tail -1 $O/mid103.txt
DIMS=ext timeout -k 10 600 python tools/sweep_regret.py 104 140 > $O/ext104.txt 2>&1; echo "ext rc=$?"; tail -1 $O/ext104.txt
NOWS=1 timeout -k 10 600 python tools/sweep_regret.py 105 100 > $O/nows105.txt 2>&1; echo "nows rc=$?"; tail -1 $O/nows105.txt
for f in $O/*.txt; do echo "== $f"; grep "^#  " $f | head -4; done
