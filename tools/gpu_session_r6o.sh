#!/bin/bash
# round 6, GPU session O: where does the LK difference beside a dense bf16 matrix-product co-runner come from?  CU-masked streams (same CU vs chip level), dose, ORB as the victim,
# pyramid bytes of the contaminated runs; LK variants (arguments: the variant names under sg_slam_amd/ab/)
set -u
O=gpurun_out/r6o; mkdir -p $O
run() { echo "$* :: $(env "$@" timeout 200 python tools/diag_lk_where.py 20 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')" | tee -a $O/where.txt; }
if [ $# -eq 0 ]; then
  run MASK=none VICTIM=lk; run MASK=disjoint VICTIM=lk; run MASK=same VICTIM=lk; run MASK=none VICTIM=orb; run MASK=same VICTIM=orb
  run MASK=none VICTIM=lk EXT_BLOCKS=256; run MASK=none VICTIM=lk EXT_BLOCKS=64; run MASK=none VICTIM=lk EXT_KIND=32
else
  for v in "$@"; do run VICTIM=lk LIB=sg_slam_amd/ab/libsgx_$v.so; done
fi
