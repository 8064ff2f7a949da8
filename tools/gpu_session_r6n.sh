#!/bin/bash
# round 6, GPU session N: LK variants beside the dense bf16 matrix-product co-runner (k_corun kind 2): how many keypoints of ~2000 differ per run
set -u
O=gpurun_out/r6n; mkdir -p $O
for l in lkbase lkdpp32 lknops32 lkboth32; do
  echo "$l: $(LKRUNS=1 CORUN=ext EXT_KIND=2 timeout 120 python tools/diag_lk_repeat.py 20 sg_slam_amd/ab/libsgx_$l.so 2>&1 | grep -v amdgpu.ids | grep -E '^rep ' | awk '{s+=$6; n++} END {print n, "runs differ,", s, "keys in total"}')" | tee -a $O/variants.txt
done
