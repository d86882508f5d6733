#!/bin/bash
# SQ-block counter study of one GEMM workload (the TA/TCP/TCC counter sets abort rocprofv3 7.2 on gfx950; each pass
# takes ~1.5 min because the workload builds > 256 MiB of rotating weights) on the GPU box (separate --pmc passes, nothing but counters in each run).
#   tools/pmc_gemm.sh <gemm|flux> <tag> [kernel_id]     -> gpurun_out/pmc_<tag>/summary.txt
set -o pipefail
w=${1:-gemm}; tag=${2:-x}; kid=${3:-0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$tag; rm -rf $O; mkdir -p $O; cd $R
passes=(
 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"
 "SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
)
i=0
for p in "${passes[@]}"; do
  timeout -k 10 200 rocprofv3 --pmc $p --output-format csv -d $O/p$i -o p -- python tools/run_workload.py $w 6 $kid > $O/p$i.log 2>&1 || { echo "pass $i failed: $p"; tail -3 $O/p$i.log; }
  i=$((i+1))
done
python tools/summarize_prof.py $O gemm_kernel > $O/summary.txt
cat $O/summary.txt
