#!/bin/bash
# round 6, GPU session A: first device run of k_hrb (sgx_det_hrb.h).  Per-step detector table with k_hrb and with k_fused_block2 (SGX_DET_HRB=0, tap build) on the same box,
# the detector GPU tests (per-step isolation, row identity), a short pipeline bench both ways.
set -u
O=gpurun_out/r6a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/prof_det_ops.py 512 5 > $O/det_ops_hrb.txt 2> $O/det_ops_hrb.err
SGX_DET_HRB=0 timeout 300 python tools/prof_det_ops.py 512 5 > $O/det_ops_fb2.txt 2> $O/det_ops_fb2.err
head -8 $O/det_ops_hrb.txt; head -8 $O/det_ops_fb2.txt | tail -6
( time timeout 900 python -m pytest tests/test_detector_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -25 ) > $O/gpu_tests_detector.log 2>&1
tail -12 $O/gpu_tests_detector.log
Q="--no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 60 --warmup 8"
timeout 300 python bench.py $Q > $O/bench_hrb.json 2> $O/bench.err
SGX_BENCH_TAPS_LIB=1 SGX_DET_HRB=0 timeout 300 python bench.py $Q > $O/bench_fb2.json 2>> $O/bench.err
for f in $O/bench_hrb.json $O/bench_fb2.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); pk = j['roofline']['per_kernel']
    print(sys.argv[1].split('/')[-1], 'fps %.0f ms/step %.3f det_forward %.3f' % (j['value'], j['ms_per_step'], pk.get('det_forward', {}).get('avg_ms_per_launch', -1)))
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:300])
PY
done
