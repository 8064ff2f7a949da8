#!/bin/bash
# round 6, GPU session E: the whole -m gpu tier on the current tree (k_hrb default plan, superseded kernels out of the product, C-ABI collective), then the default bench line
set -u
O=gpurun_out/r6e; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -30 ) > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err
python - <<'PY'
import json
j = json.load(open('gpurun_out/r6e/bench_driver_form.json'))
print('driver form:', round(j['value']), 'frames/s', round(j['ms_per_step'], 3), 'ms/step; host input', j.get('value_host_input'), '; roofline', j['roofline']['kernel'], round(j['roofline']['frac'], 4), 'excluded', j['config'].get('ate_streams_excluded'))
PY
