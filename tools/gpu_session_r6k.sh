#!/bin/bash
# round 6, GPU session K: the LK kernel of this round against the one of round 5 (sg_slam_amd/ab/libsgx_r5flow.so: sgx_flow.cpp + sgx_flow_kernels.h of commit 0f9fed7) on the stand-alone harness, alternating runs on one box
set -u
O=gpurun_out/r6k; mkdir -p $O
for i in 1 2 3; do
  for v in product r5flow; do
    if [ $v = product ]; then unset SGX_BENCH_AB_LIB; else export SGX_BENCH_AB_LIB=sg_slam_amd/ab/libsgx_$v.so; fi
    echo "$v $(timeout 200 python tools/bench_flow.py --streams 512 --reps 12 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['lk_track']['ms_per_launch'],4))")" | tee -a $O/ab.txt
  done
done
unset SGX_BENCH_AB_LIB

