#!/bin/bash
set -u
O=gpurun_out/r6i; mkdir -p $O
echo "== LK alone beside the LDS polluter"; POLLUTE=64 timeout 600 python tools/diag_lk_repeat.py 300 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/lk_pollute.txt
echo "== two trackers, kernels serialised"; AMD_SERIALIZE_KERNEL=3 timeout 900 python tools/diag_two_trackers.py 60 2>&1 | grep -v amdgpu.ids | grep -E "^rep|^reps" | tail -5 | tee $O/two_serial.txt
