#!/bin/bash
set -u
TAG=${1:-trip6}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for p in 2 4 6 0 5 7; do
  SGX_TRK_PRIO=$p timeout 300 python bench.py --no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 80 --warmup 6 > $O/prio_$p.json 2>/dev/null
  python - <<PY
import json
j = json.load(open("$O/prio_$p.json")); pk = j['roofline']['per_kernel']
print('prio', $p, round(j['value']), 'fps', round(j['ms_per_step'], 3), 'ms; det', pk['det_forward']['avg_ms_per_launch'], 'lk', pk['lk_track']['avg_ms_per_launch'], 'pose', pk['pose_opt']['avg_ms_per_launch'], 'match', pk['match_project_frame']['avg_ms_per_launch'])
PY
done
