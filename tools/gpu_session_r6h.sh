#!/bin/bash
# round 6, GPU session H: k_conv_dw3 (channel-pair depthwise) against k_conv_dw2 on the per-step harness; alternative tiles of the 5 x 5 k_hrb block; detector tests
set -u
O=gpurun_out/r6h; mkdir -p $O
for v in 1 0; do echo "== SGX_DW3=$v"; SGX_DW3=$v timeout 300 python tools/prof_det_ops.py 512 5 2>/dev/null | grep -E " dw |detector plan" | tee $O/dw3_$v.txt; done
for p in 0 3 4; do echo "== SGX_HRB_PICK=$p"; SGX_HRB_PICK=$p timeout 300 python tools/prof_det_ops.py 512 5 2>/dev/null | grep -E "block 614" | tee -a $O/hrb_d.txt; done
timeout 900 python -m pytest tests/test_detector_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -4 | tee $O/tests.txt
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_tracker_native_gpu.py -q -p no:cacheprovider -x -k with_detector 2>&1 | tail -2 | tee -a $O/native_det.txt; done
