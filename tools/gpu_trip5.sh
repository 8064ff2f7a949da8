#!/bin/bash
set -u
TAG=${1:-trip5}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
SGX_DET_GEMM=f32 timeout 200 python tools/prof_det_ops.py 512 3 2>/dev/null | grep " irb " | awk '{printf "f32      %s ms  %s %s %s\n", $1, $5, $6, $7}' >> $O/irb3_variants.txt
for d in 0 64; do
  SGX_DET_IRB3=1 SGX_IRB3_DBG=$d SGX_DET_GEMM=bf16x3 timeout 200 python tools/prof_det_ops.py 512 3 2>/dev/null | grep " irb " | awk -v d=$d '{printf "irb3 %3d %s ms  %s %s %s\n", d, $1, $5, $6, $7}' >> $O/irb3_variants.txt
done
sort -k5,5 -k1,1 $O/irb3_variants.txt
