#!/bin/bash
set -u
O=gpurun_out/r6j; mkdir -p $O
echo "== MODE=prio, eight wait states in front of every DPP read of the LK sums"; MODE=prio timeout 900 python tools/diag_two_trackers.py 200 sg_slam_amd/ab/libsgx_lkdppnops.so 2>&1 | grep -v amdgpu.ids | grep -E "^reps" | tee $O/two_prio_dppnops.txt
