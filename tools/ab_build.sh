#!/bin/bash
# A/B builds of the product library with a compile-time switch: tools/ab_build.sh NAME "-DFLAG ..."  ->  gpurun_out/ab/libsgx_NAME.so (travels to the GPU box with the snapshot? no:
# gpurun_out/ is not shipped — the variant goes to sg_slam_amd/ab/, which is git-ignored as *.so)
set -e
N=$1; F=$2; R=$(cd $(dirname $0)/.. && pwd); mkdir -p $R/sg_slam_amd/ab $R/sg_slam_amd/csrc/build_ab_$N
cd $R/sg_slam_amd/csrc
for f in sgx_*.cpp; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-everything $F -x hip -c $f -o build_ab_$N/${f%.cpp}.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared build_ab_$N/*.o -o $R/sg_slam_amd/ab/libsgx_$N.so
rm -rf build_ab_$N
echo built $R/sg_slam_amd/ab/libsgx_$N.so
