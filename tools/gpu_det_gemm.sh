#!/bin/bash
# One gpurun session of the round-4 detector work: parity of the bf16x3 plan, blob-by-blob diagnostic against the exact-fp32 plan, per-step timings of both plans, a short bench of both.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_det_gemm.sh <tag>'
set -u
TAG=${1:-trip}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 300 python tools/diag_gemm.py 3 > $O/diag_gemm.txt 2>&1
timeout 600 python -m pytest tests/test_detector_gpu.py -q -m gpu -s > $O/pytest_detector.txt 2>&1
for g in f32 bf16x3; do
  SGX_DET_GEMM=$g timeout 200 python tools/prof_det_ops.py 512 5 > $O/detector_ops_$g.txt 2>$O/detector_ops_$g.err
  SGX_BENCH_TAPS_LIB=1 SGX_DET_GEMM=$g timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-config2 --no-config4 --no-host-input > $O/bench_$g.json 2>$O/bench_$g.err
done
tail -3 $O/pytest_detector.txt; tail -4 $O/diag_gemm.txt; head -1 $O/detector_ops_f32.txt; head -1 $O/detector_ops_bf16x3.txt
python - <<PY
import json
for g in ('f32', 'bf16x3'):
    try:
        j = json.load(open("$O/bench_%s.json" % g)); print(g, round(j['value']), 'frames/s', round(j['ms_per_step'], 3), 'ms/step')
    except Exception as e: print(g, 'bench failed', e)
PY
