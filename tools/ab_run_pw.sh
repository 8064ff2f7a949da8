#!/bin/bash
# A/B on one box: the detector's pointwise steps (and the plan's sum) with the product library and the given sg_slam_amd/ab/ variants
set -u
cp sg_slam_amd/libsgx.so /tmp/libsgx_product.so
run() { timeout 200 python tools/prof_det_ops.py 512 3 2>/dev/null > /tmp/ops_$1.txt; awk -v n=$1 '/ pw /{s+=$1} /^detector/{t=$10} END {print n, "sum of pw steps", s, "ms; plan", t, "ms"}' /tmp/ops_$1.txt; }
for rep in 1 2; do
  cp /tmp/libsgx_product.so sg_slam_amd/libsgx.so; run product
  for n in "$@"; do cp sg_slam_amd/ab/libsgx_$n.so sg_slam_amd/libsgx.so; run $n; done
done
cp /tmp/libsgx_product.so sg_slam_amd/libsgx.so
paste <(grep " pw " /tmp/ops_product.txt | awk '{print $1}') <(grep " pw " /tmp/ops_$1.txt | awk '{print $1, $5, $6, $9}') | awk '{printf "%s %s  %s %s %s\n", $1, $2, $3, $4, $5}'
