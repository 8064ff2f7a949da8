#!/bin/bash
set -u
O=gpurun_out/r6i; mkdir -p $O
echo "== two trackers, group sums through v_permlane16_swap"; timeout 900 python tools/diag_two_trackers.py 160 sg_slam_amd/ab/libsgx_lkswap16.so 2>&1 | grep -v amdgpu.ids | grep -E "^reps" | tail -3 | tee $O/two_lkswap16.txt
timeout 300 python -m pytest tests/test_flow_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -2
