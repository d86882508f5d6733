#!/bin/bash
# more measured shapes for the dispatch fit: standard seeds, extended (small / ragged) dimensions, fp32 output, no workspace
set -uo pipefail
O=gpurun_out/r4s; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for seed in 51 52 53 54 55 56; do timeout -k 10 600 python tools/sweep_regret.py $seed 140 > $O/regret$seed.txt 2>&1; echo "seed $seed rc=$? $(tail -1 $O/regret$seed.txt)"; done
for seed in 57 58 59; do DIMS=ext timeout -k 10 600 python tools/sweep_regret.py $seed 140 > $O/regret${seed}_ext.txt 2>&1; echo "ext $seed rc=$? $(tail -1 $O/regret${seed}_ext.txt)"; done
OUT=f32 DIMS=ext timeout -k 10 600 python tools/sweep_regret.py 60 100 > $O/regret60_ext_f32.txt 2>&1; echo "ext f32 rc=$? $(tail -1 $O/regret60_ext_f32.txt)"
OUT=f32 timeout -k 10 600 python tools/sweep_regret.py 61 100 > $O/regret61_f32.txt 2>&1; echo "f32 rc=$? $(tail -1 $O/regret61_f32.txt)"
for seed in 62 63; do NOWS=1 timeout -k 10 600 python tools/sweep_regret.py $seed 120 > $O/regret${seed}_nows.txt 2>&1; echo "nows $seed rc=$? $(tail -1 $O/regret${seed}_nows.txt)"; done
NOWS=1 DIMS=ext timeout -k 10 600 python tools/sweep_regret.py 64 100 > $O/regret64_ext_nows.txt 2>&1; echo "nows ext rc=$? $(tail -1 $O/regret64_ext_nows.txt)"
