#!/bin/bash
# round 5, GPU session D1: the dynamic scene at 512 streams, a few more pipeline A/Bs on the tap build (chain capture), the GPU test tier on the final tree
set -u
O=gpurun_out/r5d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 60 --warmup 8"
timeout 300 python bench.py $Q > $O/sc_layered.json 2>> $O/ab.err
timeout 300 python bench.py $Q --scene dynamic > $O/sc_dynamic.json 2>> $O/ab.err
timeout 300 python bench.py $Q --scene dynamic --person-logit 0 > $O/sc_dynamic_person0.json 2>> $O/ab.err
ab() { name=$1; shift; env SGX_BENCH_TAPS_LIB=1 "$@" timeout 300 python bench.py $Q > $O/ab_$name.json 2>> $O/ab.err; }
ab base X=1
ab lk_kpw4 SGX_LK_KPW=4
ab prio0 SGX_TRK_PRIO=0
ab prio1 SGX_TRK_PRIO=1
ab prio4 SGX_TRK_PRIO=4
ab prio7 SGX_TRK_PRIO=7
for f in $O/sc_*.json $O/ab_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); c = j['config']; pk = j['roofline']['per_kernel']
    g = lambda k: pk.get(k, {}).get('avg_ms_per_launch', -1)
    print('%-26s fps %.0f ms/step %.3f tracked %s kp %.0f->%.0f match %.0f inl %.0f ransac_it %.1f boxes %.2f ate_gt %.4f | det %.2f lk %.2f ransac %.2f' % (sys.argv[1].split('/')[-1], j['value'], j['ms_per_step'], c['tracked_streams_last_frame'], c['mean_keypoints_before_mask'], c['mean_keypoints'], c['mean_matches'], c['mean_inliers'], c['mean_ransac_iterations'], c['detector']['mean_person_boxes_last_step'], c['ate_rmse_m_vs_ground_truth'], g('det_forward'), g('lk_track'), g('fm_ransac')))
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:200])
PY
done > $O/summary.txt 2>&1
cat $O/summary.txt
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 ) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
