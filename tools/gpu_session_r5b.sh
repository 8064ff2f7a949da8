#!/bin/bash
# round 5, GPU session B: the GPU test tier on the tree with the forked detector graph on by default, A/B of the remaining occupancy taps inside the pipeline (tap build),
# then the full profile collection (tools/collect_profiles.sh r5).  Outputs under gpurun_out/r5b/ and gpurun_out/r5/.
set -u
O=gpurun_out/r5b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 ) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
Q="--no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 60 --warmup 8"
ab() { name=$1; shift; env SGX_BENCH_TAPS_LIB=1 "$@" timeout 300 python bench.py $Q > $O/ab_$name.json 2>> $O/ab.err; }
ab base_taps_1
ab irb_split300 SGX_IRB_SPLIT=300
ab irb_nbuf1 SGX_IRB_NBUF=1
ab match512 SGX_TUNE_MATCH_THREADS=512
ab match256 SGX_TUNE_MATCH_THREADS=256
ab fork1 SGX_DET_FORK=1
ab base_taps_2
for f in $O/ab_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); c = j['config']; pk = (j.get('roofline') or {}).get('per_kernel', {})
    g = lambda k: pk.get(k, {}).get('avg_ms_per_launch', -1)
    print('%-28s fps %.0f ms/step %.3f tracked %s det_fwd %.3f mpf %.3f mpl %.3f lk %.3f' % (sys.argv[1].split('/')[-1], j['value'], j['ms_per_step'], c['tracked_streams_last_frame'], g('det_forward'), g('match_project_frame'), g('match_project_local'), g('lk_track')))
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:200])
PY
done > $O/summary.txt 2>&1
cat $O/summary.txt
bash tools/collect_profiles.sh r5 2>&1 | tail -5
