#!/bin/bash
# tools/r4_cold.sh <part>: the measured times the dispatch is fitted on, re-taken with COLD weights (>= 320 MiB in rotation for every shape)
set -uo pipefail
O=gpurun_out/r4c2; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
run() { # name env... -- seed count
  local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 1100 python tools/sweep_regret.py $1 $2 > $O/$name.txt 2>&1; echo "$name rc=$? $(tail -1 $O/$name.txt)"
}
case ${1:-a} in
 a) for s in 101 102 103 104 105 106; do run std$s -- $s 140; done
    run ext113 DIMS=ext -- 113 130; run ext114 DIMS=ext -- 114 130 ;;
 b) for s in 107 108 109 110; do run std$s -- $s 140; done
    run midm117 MS=200,224,256,288,320,384,448,512,640,768,896,1024 -- 117 130; run midm118 MS=200,256,300,320,384,400,512,600,768,1000,1024 DIMS=ext -- 118 120
    run f32_121 OUT=f32 -- 121 110; run ext_f32_122 OUT=f32 DIMS=ext -- 122 90 ;;
 c) F32="512,4096,4096;1024,4096,4096;512,4096,8192;512,8192,4096;384,4096,4096;640,4096,4096;512,3072,4096;768,4096,4096;256,4096,4096;512,4096,3072"
    BF="4096,3072,12288;4096,3072,1536;1536,3072,4096;2048,4096,4096;64,14336,4096;4,4096,4096;512,4096,4096;1024,4096,4096;256,4096,4096;4096,4096,4096;8192,8192,8192;4096,3072,3072;128,4096,4096;32,14336,4096;2,4096,4096;8,4096,4096;512,4096,8192;768,3072,3072"
    for r in 1 2 3; do run anchors_f32_$r OUT=f32 SHAPES="$F32" -- 9$r 10; run anchors_bf16_$r SHAPES="$BF" -- 9$r 18; done
    for s in 127 128 129 130; do run heldout$s -- $s 140; done
    run heldout_ext131 DIMS=ext -- 131 120; run heldout_midm132 MS=200,256,320,384,448,512,640,768,1024 -- 132 110
    run nows123 NOWS=1 -- 123 120; run nows124 NOWS=1 -- 124 120; run nows_ext125 NOWS=1 DIMS=ext -- 125 100; run nows_midm126 NOWS=1 MS=200,256,320,384,512,768,1024 -- 126 90 ;;
 d) for s in 133 134 135 136 137 138 139 140 141 142; do run std$s -- $s 140; done
    for s in 143 144 145; do run ext$s DIMS=ext -- $s 130; done
    run midm146 MS=200,224,256,288,320,384,448,512,640,768,896,1024 -- 146 130; run midm147 MS=200,256,300,320,384,400,512,600,768,1000,1024 DIMS=ext -- 147 120; run midm148 MS=256,384,512,768,1024 -- 148 120
    run f32_149 OUT=f32 -- 149 120; run f32_150 OUT=f32 MS=200,256,384,512,768,1024 -- 150 100 ;;
esac
