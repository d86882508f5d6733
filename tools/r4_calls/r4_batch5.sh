#!/bin/bash
set -uo pipefail
O=gpurun_out/r4e; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for seed in 35 36; do
  timeout -k 10 900 python tools/sweep_regret.py $seed 140 > $O/regret$seed.txt 2>&1; echo "regret $seed rc=$?"; tail -1 $O/regret$seed.txt
done
NOWS=1 timeout -k 10 600 python tools/sweep_regret.py 37 60 > $O/regret37_nows.txt 2>&1; echo "regret nows rc=$?"; tail -1 $O/regret37_nows.txt
for f in $O/regret3*.txt; do echo "== $f"; grep "^#  " $f | head -6; done
timeout -k 10 900 python bench.py --steps 10 --warmup 3 --no-host-kernarg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4e/bench.json'))
print('value', d['value'], d['unit'], 'kernel_avg_us', d['roofline']['kernel_avg_us'], 'frac', d['roofline']['frac'])
s=d.get('secondary',{})
for k in ('gemv','gemv_sq','flux','mid','wide','skinny','decode','linear','quantize','dequant'):
    r=(s.get(k,{}).get('roofline') or {}); print(k, s.get(k,{}).get('value'), r.get('kernel_avg_us'), r.get('frac'))
PY
