#!/bin/bash
set -uo pipefail
O=gpurun_out/r4x; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for seed in 46 47 48 49; do timeout -k 10 600 python tools/fuzz_mm.py $seed 300 > $O/fuzz_mm$seed.txt 2>&1; echo "fuzz_mm $seed rc=$? $(tail -1 $O/fuzz_mm$seed.txt)"; done
timeout -k 10 600 python tools/fuzz_large.py 50 36 > $O/fuzz_large.txt 2>&1; echo "fuzz_large rc=$? $(tail -1 $O/fuzz_large.txt)"
timeout -k 10 300 python tools/fuzz_casts.py 45 150 > $O/fuzz_casts.txt 2>&1; echo "fuzz_casts rc=$? $(tail -1 $O/fuzz_casts.txt)"
timeout -k 10 300 python tools/check_splitk.py > $O/splitk.txt 2>&1; echo "check_splitk rc=$? $(tail -1 $O/splitk.txt)"
timeout -k 10 300 python bench.py --workload gemv_c1 --steps 8 --warmup 3 --no-cpu-baseline --no-ceilings > $O/c1.json 2> $O/c1.err; echo "c1 rc=$?"; python -c "import json;d=json.load(open('$O/c1.json'));print(d['value'],d['unit'],d['roofline']['kernel_avg_us'],d['roofline']['frac'])"
