#!/bin/bash
set -u
TAG=${1:-trip9}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for a in 0 1; do
  SGX_DET_IRB_A3=$a timeout 200 python tools/prof_det_ops.py 512 3 2>/dev/null | grep -E " irb |^detector" | awk -v a=$a '{printf "A3=%d  %s ms  %s %s %s\n", a, $1, $5, $6, $7}' >> $O/irb_a3.txt
  SGX_DET_IRB_A3=$a timeout 300 python bench.py --no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 80 --warmup 6 > $O/bench_a3_$a.json 2>/dev/null
  python -c "
import json; j=json.load(open('$O/bench_a3_$a.json')); print('A3=$a bench', round(j['value']), 'fps', round(j['ms_per_step'],3), 'ms')" >> $O/irb_a3.txt
done
SGX_DET_IRB_A3=1 timeout 300 python tools/diag_gemm.py 2 2>&1 | grep -E "^==|worst" >> $O/irb_a3.txt
sort -s -k5,5 $O/irb_a3.txt
