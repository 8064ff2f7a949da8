#!/bin/bash
# A/B build of the TAP library that differs in one translation unit: tools/ab_build_taps.sh NAME sgx_flow.cpp "-DFLAG ..."  ->  sg_slam_amd/ab/libsgx_NAME.so (make -C sg_slam_amd/csrc taps first)
set -e
N=$1; TU=$2; F=$3; R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/sg_slam_amd/ab
cd $R/sg_slam_amd/csrc; B=${TU%.cpp}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-everything -DSGX_DEBUG_TAPS $F -x hip -c $TU -o build/ab_${N}_taps_$B.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $(ls build/taps_*.o | grep -v build/taps_$B.o) build/ab_${N}_taps_$B.o -o $R/sg_slam_amd/ab/libsgx_$N.so
echo built $R/sg_slam_amd/ab/libsgx_$N.so
