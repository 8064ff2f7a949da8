#!/bin/bash
# timing taps of k_irb3 (SGX_IRB3_DBG bit mask; results are wrong by design): where does the block's time go?
set -u
TAG=${1:-trip3}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for d in 0 1 2 3 4 8 12 16 32 48 19 35; do
  SGX_IRB3_DBG=$d SGX_DET_GEMM=bf16x3 timeout 200 python tools/prof_det_ops.py 512 3 2>/dev/null | grep " irb " | awk -v d=$d '{printf "dbg %2d  %s ms  %s %s %s\n", d, $1, $5, $6, $7}' >> $O/irb3_taps.txt
done
cat $O/irb3_taps.txt | grep -E "c112->672->112|c80->200->80|c672->672->84"
