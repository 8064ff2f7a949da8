#!/bin/bash
# A/B on one box: DetectionOutput stand-alone (tools/prof_det_output.py) with the product library and the given sg_slam_amd/ab/ variants
set -u
cp sg_slam_amd/libsgx.so /tmp/libsgx_product.so
run() { timeout 200 python tools/prof_det_output.py 512 10 2>/dev/null | grep -E "ms per launch" | tr '\n' ' ' | awk -v n=$1 '{print n, $0}'; }
for rep in 1 2 3; do
  cp /tmp/libsgx_product.so sg_slam_amd/libsgx.so; run product
  for n in "$@"; do cp sg_slam_amd/ab/libsgx_$n.so sg_slam_amd/libsgx.so; run $n; done
done
cp /tmp/libsgx_product.so sg_slam_amd/libsgx.so
