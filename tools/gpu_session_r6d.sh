#!/bin/bash
# round 6, GPU session D: A/B of k_hrb's compile-time switches (tools/ab_build_det.sh variants swapped in as the tap library) on the per-step harness
set -u
O=gpurun_out/r6d; mkdir -p $O
cp tests/taps/libsgx_taps.so /tmp/taps_orig.so
for n in "$@"; do cp sg_slam_amd/ab/libsgx_$n.so tests/taps/libsgx_taps.so; timeout 300 python tools/prof_det_ops.py 512 5 2>/dev/null | head -7 | tail -4 | awk -v n=$n '{printf "%-8s %s  %s %s %s\n", n, $1, $5, $6, $11}' | tee -a $O/ab.txt; done
cp /tmp/taps_orig.so tests/taps/libsgx_taps.so
