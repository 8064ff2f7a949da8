#!/bin/bash
set -u
TAG=${1:-trip8}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
for v in 0 1 2; do
  SGX_TRK_SHARE=$v timeout 300 python bench.py --no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 80 --warmup 6 > $O/share_$v.json 2>/dev/null
  python - <<PY
import json
j = json.load(open("$O/share_$v.json")); pk = j['roofline']['per_kernel']
print('share', $v, round(j['value']), 'fps', round(j['ms_per_step'], 3), 'ms; det', pk['det_forward']['avg_ms_per_launch'], 'pose', pk['pose_opt']['avg_ms_per_launch'], 'match', pk['match_project_frame']['avg_ms_per_launch'], 'local', pk['match_project_local']['avg_ms_per_launch'], 'lk', pk['lk_track']['avg_ms_per_launch'], 'tracked', j['config']['tracked_streams_last_frame'])
PY
done
