#!/bin/bash
set -uo pipefail
O=gpurun_out/r4u; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 1000 python tools/sweep_split.py 81 60 > $O/split81.txt 2>&1; echo "rc=$?"; tail -1 $O/split81.txt; grep -c "^M=" $O/split81.txt
