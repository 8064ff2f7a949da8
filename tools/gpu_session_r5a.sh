#!/bin/bash
# round 5, GPU session A: the whole GPU test tier on the new tree (tap split, calibrated detector weights, per-step isolation), the driver's bench form, and A/B of the two
# host-side concurrency levers (--groups, forked detector graph).  Outputs under gpurun_out/r5a/.
set -u
O=gpurun_out/r5a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -s 2>&1 | tail -150 ) > $O/gpu_tests.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
Q="--no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 60 --warmup 8"
for rep in 1 2; do
  timeout 300 python bench.py $Q > $O/ab_base_$rep.json 2>> $O/ab.err
  timeout 300 python bench.py $Q --groups 2 > $O/ab_groups2_$rep.json 2>> $O/ab.err
  SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=3 timeout 300 python bench.py $Q > $O/ab_fork3_$rep.json 2>> $O/ab.err
  SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=3 timeout 300 python bench.py $Q --groups 2 > $O/ab_fork3_groups2_$rep.json 2>> $O/ab.err
done
timeout 300 python bench.py $Q --person-logit -0.5 > $O/ab_person05.json 2>> $O/ab.err
timeout 300 python bench.py $Q --groups 4 > $O/ab_groups4.json 2>> $O/ab.err
SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=1 timeout 200 python tools/bench_det.py 512,256 > $O/det_fork1.json 2>> $O/ab.err
SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=3 timeout 200 python tools/bench_det.py 512,256 > $O/det_fork3.json 2>> $O/ab.err
SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=5 timeout 200 python tools/bench_det.py 512,256 > $O/det_fork5.json 2>> $O/ab.err
for f in $O/ab_*.json $O/bench_driver.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); c = j['config']
    pk = (j.get('roofline') or {}).get('per_kernel', {})
    print(sys.argv[1].split('/')[-1], 'fps %.0f ms/step %.3f tracked %s boxes %.2f det_fwd %.3f' % (j['value'], j['ms_per_step'], c['tracked_streams_last_frame'], (c['detector'] or {}).get('mean_person_boxes_last_step', -1), pk.get('det_forward', {}).get('avg_ms_per_launch', -1)))
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:200])
PY
done > $O/summary.txt 2>&1
cat $O/summary.txt
tail -5 $O/gpu_tests.log
