#!/bin/bash
# round 6, GPU session G: k_irb with its expand stage as bf16x3 — in-kernel split (SGX_DET_IRB_A3=1) against pre-split operands (=2), per-step times and per-step isolation
set -u
O=gpurun_out/r6g; mkdir -p $O
for a in 0 1 2; do echo "== A3=$a"; SGX_DET_IRB_A3=$a timeout 300 python tools/prof_det_ops.py 512 5 2>/dev/null | grep -E "irb .*c(40|80|112)->|detector plan" | tee $O/a3_$a.txt; done
SGX_DET_IRB_A3=2 timeout 600 python -m pytest tests/test_detector_gpu.py -q -p no:cacheprovider -k "every_plan_step or rows_identical_to_oracle_on_8" 2>&1 | tail -4 | tee $O/tests_a3s.txt
