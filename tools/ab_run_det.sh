#!/bin/bash
# A/B on one box: per-step detector timings (tools/prof_det_ops.py 512 3) with the product library and with each sg_slam_amd/ab/libsgx_<name>.so swapped in; rows matching $PAT
set -u
R=$PWD; PAT=${PAT:-" irb |^detector"}
cp sg_slam_amd/libsgx.so /tmp/libsgx_product.so
run() { timeout 200 python tools/prof_det_ops.py 512 3 2>/dev/null | grep -E "$PAT" | awk -v n=$1 '{printf "%-10s %s ms  %s %s %s %s\n", n, $1, $4, $5, $6, $7}'; }
for rep in $(seq ${REPS:-2}); do
  cp /tmp/libsgx_product.so sg_slam_amd/libsgx.so; run product
  for n in "$@"; do cp sg_slam_amd/ab/libsgx_$n.so sg_slam_amd/libsgx.so; run $n; done
done
cp /tmp/libsgx_product.so sg_slam_amd/libsgx.so
