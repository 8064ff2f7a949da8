#!/bin/bash
# A/B on one box: the chain on one stream (standalone kernel times) with the product library and with each sg_slam_amd/ab/libsgx_<name>.so swapped in; 3 alternations
set -u
R=$PWD; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; shift
cp sg_slam_amd/libsgx.so /tmp/libsgx_product.so
run() { python bench.py --no-cpu-baseline --no-detector --no-config2 --no-config4 --no-host-input --no-pipeline --steps 48 --warmup 4 2>/dev/null | python -c "
import json,sys; j=json.load(sys.stdin); pk=j['roofline']['per_kernel']; print('$1', ' '.join('%s %.4f' % (k, pk[k]['avg_ms_per_launch']) for k in ('pyramid_resize','fast_cells','octree','orient_desc','lk_track','lk_pyramid','pose_opt','fm_ransac')))"; }
for rep in 1 2 3; do
  cp /tmp/libsgx_product.so sg_slam_amd/libsgx.so; run product
  for n in "$@"; do cp sg_slam_amd/ab/libsgx_$n.so sg_slam_amd/libsgx.so; run $n; done
done
cp /tmp/libsgx_product.so sg_slam_amd/libsgx.so
