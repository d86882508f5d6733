#!/bin/bash
# quick bench line: tools/bq.sh <workload> <kernel> [extra bench args]  -> "workload kernel value avg_us frac"
w=$1; k=$2; shift 2
timeout -k 10 120 python bench.py --workload $w --kernel $k --steps 8 --warmup 3 --no-cpu-baseline "$@" > /tmp/bq.json 2>/tmp/bq.err || { echo "$w k=$k FAILED"; tail -3 /tmp/bq.err; exit 0; }
python -c "import json;d=json.load(open('/tmp/bq.json'));r=d['roofline'];print('$w k=$k $*', 'value', d['value'], d['unit'], 'kernel_avg_us', r['kernel_avg_us'], 'min', r['kernel_min_us'], 'frac', r['frac'])"
