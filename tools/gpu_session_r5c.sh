#!/bin/bash
# round 5, GPU session C: the host-input leg against the forked detector graph (one / two executable instances), the GPU test tier, device-side differential campaigns
set -u
O=gpurun_out/r5c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--no-cpu-baseline --no-config2 --no-config4 --steps 40 --warmup 8 --host-steps 60"
ab() { name=$1; shift; env "$@" timeout 300 python bench.py $Q > $O/hi_$name.json 2>> $O/ab.err; }
ab product_1 X=1
ab fork1 SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=1
ab fork3_exec1 SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=3 SGX_DET_EXECS=1
ab fork3_exec2 SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=3 SGX_DET_EXECS=2
ab fork1_exec2 SGX_BENCH_TAPS_LIB=1 SGX_DET_FORK=1 SGX_DET_EXECS=2
ab product_2 X=1
for f in $O/hi_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); h = j['host_input']; pk = j['roofline']['per_kernel']
    print('%-22s fps %.0f ms/step %.3f | host-input fps %.0f ms/step %.3f upload %.1f GB/s | det_fwd %.2f mpf %.2f' % (sys.argv[1].split('/')[-1], j['value'], j['ms_per_step'], h['value'], h['ms_per_step'], h['upload_GBs'], pk['det_forward']['avg_ms_per_launch'], pk['match_project_frame']['avg_ms_per_launch']))
except Exception as e:
    print(sys.argv[1], 'FAILED', repr(e)[:200])
PY
done > $O/summary.txt 2>&1
cat $O/summary.txt
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 ) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
bash tools/gpu_campaigns.sh r5c_camp 2>&1 | tail -8
