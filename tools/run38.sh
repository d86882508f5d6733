#!/bin/bash
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd "$R"
bash tools/r2_profiles.sh > gpurun_out/r2_profiles_final.log 2>&1; echo "profiles rc=$?"; tail -5 gpurun_out/r2_profiles_final.log
cp gpurun_out/prof_r02/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
python bench.py > gpurun_out/bench_final2.json 2> gpurun_out/bench_final2.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_final2.json
