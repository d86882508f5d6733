#!/bin/bash
# round-3 profiles in one GPU call: kernel trace of the default bench + FETCH/WRITE per workload, then the counter studies of C3, FLUX, decode and the vec-mat
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd "$R"
bash tools/profile_round.sh r03 > gpurun_out/profile_r03.log 2>&1; echo "profile_round rc=$?"; tail -14 gpurun_out/profile_r03.log
for wt in "gemm c3_r03" "flux flux_r03" "decode decode_r03"; do set -- $wt; bash tools/pmc_gemm.sh $1 $2 > gpurun_out/pmc_$2.log 2>&1; echo "pmc $1 rc=$? failed passes: $(grep -c failed gpurun_out/pmc_$2.log)"; done
