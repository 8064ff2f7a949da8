#!/bin/bash
set -u
TAG=${1:-trip7}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for v in "0 0" "1 0" "1 64"; do
  set -- $v
  SGX_DET_IRB3=$1 SGX_IRB3_DBG=$2 timeout 300 python bench.py --no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 80 --warmup 6 > $O/irb3_$1_$2.json 2>/dev/null
  python - <<PY
import json
j = json.load(open("$O/irb3_$1_$2.json")); pk = j['roofline']['per_kernel']
print('irb3', $1, 'dbg', $2, round(j['value']), 'fps', round(j['ms_per_step'], 3), 'ms; det in-pipeline', pk['det_forward']['avg_ms_per_launch'])
PY
done
