#!/bin/bash
# gpurun session: split-cost micro-benchmark + per-step PMC tables of the detector plan with exact-fp32 and bf16x3 matrix products
set -u
TAG=${1:-trip2}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 60 tools/ubench/bf16_split > $O/ubench_bf16_split.txt 2>&1
for g in f32 bf16x3; do
  SGX_DET_GEMM=$g bash tools/pmc_irb.sh ${TAG}_$g > /dev/null 2>&1
  cp gpurun_out/pmc_irb_${TAG}_$g.txt $O/pmc_steps_$g.txt
done
cat $O/ubench_bf16_split.txt
