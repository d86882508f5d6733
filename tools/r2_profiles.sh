#!/bin/bash
# round-2 profiles: kernel trace of the default bench + FETCH/WRITE per workload, then the counter study of C3 and FLUX
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd "$R"
bash tools/profile_round.sh r02 > gpurun_out/profile_r02.log 2>&1; echo "profile_round rc=$?"; tail -25 gpurun_out/profile_r02.log
bash tools/pmc_gemm.sh gemm c3_r02 > gpurun_out/pmc_c3_r02.log 2>&1; echo "pmc c3 rc=$?"; grep -c "failed" gpurun_out/pmc_c3_r02.log
bash tools/pmc_gemm.sh flux flux_r02 > gpurun_out/pmc_flux_r02.log 2>&1; echo "pmc flux rc=$?"; grep -c "failed" gpurun_out/pmc_flux_r02.log
