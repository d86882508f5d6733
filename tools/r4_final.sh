#!/bin/bash
# final GPU run of round 4: the whole GPU suite, the round profile (kernel trace + FETCH / WRITE per workload -> pmc_traffic.json), then the bench line on the same sources
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd "$R"
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_final.log
bash tools/profile_round.sh r04 > gpurun_out/profile_r04.log 2>&1; echo "profile_round rc=$?"; tail -14 gpurun_out/profile_r04.log
cp gpurun_out/prof_r04/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r04.json'))
r=d['roofline']; print('value',d['value'],d['unit'],'kernel',r['kernel_avg_us'],'frac',r['frac'],'traffic',r['traffic'],'floor',{k:v for k,v in (r.get('floor') or {}).items() if k.endswith('_us') or k.startswith('kernel_over')})
for k,v in d['secondary'].items():
    if isinstance(v,dict) and v.get('roofline'): print(k, v['value'], v['unit'], v['roofline']['kernel_avg_us'], v['roofline']['frac'], v['roofline'].get('traffic'), (v['roofline'].get('floor') or {}).get('dma_only_us'))
PY
