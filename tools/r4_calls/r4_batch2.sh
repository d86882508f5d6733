#!/bin/bash
set -uo pipefail
O=gpurun_out/r4b; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python tools/time_unaligned.py > $O/unaligned.txt 2>&1; echo "unaligned rc=$?"; cat $O/unaligned.txt
timeout -k 10 300 tools/probes/mlp_probe > $O/mlp_probe.txt 2>&1; echo "mlp rc=$?"; cat $O/mlp_probe.txt
timeout -k 10 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4b/bench.json'))
print('value', d['value'], d['unit'], 'roofline', {k:v for k,v in d['roofline'].items() if k in ('frac','kernel_avg_us','floor')})
s=d.get('secondary',{})
print('host_kernarg', s.get('gemm_host_kernarg'))
print('wide floor', (s.get('wide',{}).get('roofline') or {}).get('floor'))
for k in ('gemv','flux','mid','wide','skinny','decode'):
    r=(s.get(k,{}).get('roofline') or {}); print(k, r.get('kernel_avg_us'), r.get('frac'))
PY
