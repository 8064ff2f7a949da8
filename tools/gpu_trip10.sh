#!/bin/bash
set -u
TAG=${1:-trip10}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do for m in 64 0; do
  SGX_PW3_MINK=$m timeout 300 python bench.py --no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 80 --warmup 6 > $O/mink_$m.json 2>/dev/null
  python -c "
import json; j=json.load(open('$O/mink_$m.json')); print('mink $m', round(j['value']), 'fps', round(j['ms_per_step'],3), 'ms; det in-pipeline', j['roofline']['per_kernel']['det_forward']['avg_ms_per_launch'])"
done; done
