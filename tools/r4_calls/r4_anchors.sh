#!/bin/bash
# the BASELINE configurations, bench.py's workloads and their neighbours against every kernel that takes them (anchors of the dispatch fit), three repeats each
set -uo pipefail
O=gpurun_out/r4w; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
F32="512,4096,4096;1024,4096,4096;512,4096,8192;512,8192,4096;384,4096,4096;640,4096,4096;512,3072,4096;768,4096,4096;256,4096,4096;512,4096,3072"
BF="4096,3072,12288;4096,3072,1536;1536,3072,4096;2048,4096,4096;64,14336,4096;4,4096,4096;512,4096,4096;1024,4096,4096;256,4096,4096;4096,4096,4096;8192,8192,8192;4096,3072,3072;128,4096,4096;32,14336,4096;2,4096,4096;8,4096,4096;512,4096,8192;768,3072,3072"
for r in 1 2 3; do
  OUT=f32 SHAPES="$F32" timeout -k 10 600 python tools/sweep_regret.py 9$r > $O/anchors_f32_$r.txt 2>&1; echo "f32 $r rc=$? $(tail -1 $O/anchors_f32_$r.txt)"
  SHAPES="$BF" timeout -k 10 600 python tools/sweep_regret.py 9$r > $O/anchors_bf16_$r.txt 2>&1; echo "bf16 $r rc=$? $(tail -1 $O/anchors_bf16_$r.txt)"
done
grep "^M=" $O/anchors_f32_1.txt | head -4
