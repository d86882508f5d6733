#!/bin/bash
set -uo pipefail
O=gpurun_out/r4c; mkdir -p $O
MLP_QUICK=1 timeout -k 10 300 tools/probes/mlp_probe > $O/mlp_probe2.txt 2>&1; echo "mlp rc=$?"; cat $O/mlp_probe2.txt
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 120 tools/probes/kernarg_probe > $O/kernarg_dev.txt 2>&1; echo "kernarg dev rc=$?"; cat $O/kernarg_dev.txt
HIP_FORCE_DEV_KERNARG=0 timeout -k 10 120 tools/probes/kernarg_probe > $O/kernarg_host.txt 2>&1; echo "kernarg host rc=$?"; cat $O/kernarg_host.txt
