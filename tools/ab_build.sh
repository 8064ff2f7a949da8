#!/bin/bash
# A/B build of the PRODUCT library that differs in one translation unit: tools/ab_build.sh NAME sgx_flow.cpp "-DFLAG ..."  ->  sg_slam_amd/ab/libsgx_NAME.so (git-ignored, travels with
# gpurun).  The other objects are the product's (make -C sg_slam_amd/csrc ../libsgx.so first).
set -e
N=$1; TU=$2; F=$3; R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/sg_slam_amd/ab
cd $R/sg_slam_amd/csrc; B=${TU%.cpp}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-everything $F -x hip -c $TU -o build/ab_${N}_$B.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $(ls build/sgx_*.o | grep -v build/$B.o) build/ab_${N}_$B.o -o $R/sg_slam_amd/ab/libsgx_$N.so
echo built $R/sg_slam_amd/ab/libsgx_$N.so
