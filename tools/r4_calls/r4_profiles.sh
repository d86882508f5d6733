#!/bin/bash
# round-4 profiles in one GPU call: kernel trace of the default bench + FETCH/WRITE per workload (-> profiles/pmc_traffic.json), then the counter studies of C3, wide and FLUX
set -uo pipefail
R=${GRAFT_REPO_ROOT:?}; cd "$R"
bash tools/profile_round.sh r04 > gpurun_out/profile_r04.log 2>&1; echo "profile_round rc=$?"; tail -16 gpurun_out/profile_r04.log
for wt in "gemm c3_r04" "wide wide_r04" "flux flux_r04"; do set -- $wt; bash tools/pmc_gemm.sh $1 $2 > gpurun_out/pmc_$2.log 2>&1; echo "pmc $1 rc=$? failed passes: $(grep -c failed gpurun_out/pmc_$2.log)"; done
export HIP_FORCE_DEV_KERNARG=1
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err; echo "bench rc=$?"
