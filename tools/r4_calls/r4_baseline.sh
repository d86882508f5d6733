#!/bin/bash
# round-4 baseline on this round's box: GPU tests, the self-launched 2-rank gloo rehearsal, per-dispatch times of the shapes the verdict names
set -uo pipefail
O=gpurun_out/r4a; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -3 $O/pytest.log
FP8MI_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err; echo "rehearse rc=$?"
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r4a/rehearse2.json')); print('rehearse2: n_gpus', d['n_gpus'], 'value', d['value'], d['unit'], 'rccl_ranks', d.get('rccl_ranks'), 'uuids', d.get('device_uuids'))
except Exception as e: print('rehearse2 unreadable', e)
PY
for a in "512 4096 4096 0 f32" "1024 4096 4096 0 f32" "4096 3072 1536 0 bf16" "2048 4096 4096 0 bf16" "4096 3072 12288 0 bf16" "1 14336 4096 0 f32" "1 4096 4096 0 f32" "64 14336 4096 0 bf16" "4096 4100 4096 0 bf16" "4096 4096 4096 0 bf16"; do
  timeout -k 10 120 python tools/time_shape.py $a 60 2>&1 | tail -1 | tee -a $O/shapes.txt
done
