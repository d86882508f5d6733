#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; O="$R/gpurun_out/r2p9"; mkdir -p "$O"; cd "$R"
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "rne or transposed or sharded" > "$O/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$O/pytest.log"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; echo "smoke rc=$?"; tail -4 "$O/smoke.log"
( time timeout -k 10 900 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err" ) 2> "$O/bench.time"; echo "bench rc=$?"; cat "$O/bench.time"; tail -3 "$O/bench.err"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2p9/bench.json"))
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "launches/step", d["config"]["launches_per_step"])
print("roofline", d["roofline"]); print("ceilings", d.get("measured_ceilings")); print("cpu", d.get("cpu_baseline"))
for k,v in d.get("secondary",{}).items():
    if "error" in v: print(k, v); continue
    r=v.get("roofline") or {}
    print(f"{k:13s} value {v['value']:10.1f} {v['unit']:8s} ms/step {v['ms_per_step']:8.3f} launches {v['launches_per_step']:5d} kernel_us {r.get('kernel_avg_us')} frac {r.get('frac')}")
PY
