#!/bin/bash
# A/B builds of the tap library that differ only in sgx_det.cpp: tools/ab_build_det.sh NAME "-DFLAG ..."  ->  sg_slam_amd/ab/libsgx_NAME.so (git-ignored, travels with gpurun).
# The other translation units are taken from the tap build (make -C sg_slam_amd/csrc taps first).
set -e
N=$1; F=$2; R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/sg_slam_amd/ab
cd $R/sg_slam_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-everything -DSGX_DEBUG_TAPS $F -x hip -c sgx_det.cpp -o build/ab_${N}_sgx_det.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $(ls build/taps_*.o | grep -v taps_sgx_det.o) build/ab_${N}_sgx_det.o -o $R/sg_slam_amd/ab/libsgx_$N.so
echo built $R/sg_slam_amd/ab/libsgx_$N.so
