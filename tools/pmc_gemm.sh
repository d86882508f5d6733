#!/bin/bash
# Counter study of one GEMM workload on the GPU box: separate --pmc passes, nothing but counters in each run.
#   tools/pmc_gemm.sh <gemm|flux|...> <tag> [kernel_id]     -> gpurun_out/pmc_<tag>/summary.txt
#
# Why round 1's TA / TCP / TD / TCC passes ended in "rocprofv3 caught signal 6" (gpurun_out/pmc_c3.txt): two causes,
# both in the pass lists, neither in the kernels - (a) names that `rocprofv3 -L` does not list for gfx950
# (TA_BUFFER_TOTAL_CYCLES_sum, TCP_TCR_TCP_STALL_CYCLES_sum, TD_SPI_STALL_sum, TCP_TCP_LATENCY_sum, TCP_TOTAL_READ_sum ...)
# and (b) more counters of one block than it has slots per pass (MI355X_MICROARCH.md: SQ 8, TCC 4 - FETCH_SIZE costs 3 and
# WRITE_SIZE 2 of them - GRBM 2; the pass with eight TCC counters aborted, the ones with four ran).  Every pass below
# uses only listed names (checked against gpurun_out/avail.txt by tools/check_pmc_names.py), at most 8 SQ, 4 TCC and
# 2 TA / TCP / TD counters.
set -uo pipefail
w=${1:-gemm}; tag=${2:-x}; kid=${3:-0}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}; O="$R/gpurun_out/pmc_$tag"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
passes=(
 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"
 "SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum"
 "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for p in "${passes[@]}"; do
  # shellcheck disable=SC2086
  timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d "$O/p$i" -o p -- python tools/run_workload.py "$w" 6 "$kid" > "$O/p$i.log" 2>&1 || { echo "pass $i failed: $p"; tail -3 "$O/p$i.log"; }
  i=$((i+1))
done
python tools/summarize_prof.py "$O" gemm > "$O/summary.txt"   # gemm_kernel<...> (ring) or gemm256_kernel<0>
cat "$O/summary.txt"
