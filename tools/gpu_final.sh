#!/bin/bash
# last GPU trip of a round: the -m gpu suite on the final tree (log kept under profiles/), then bounded device-side differential campaigns (tools/campaign_*.py with
# SGX_CAMPAIGN_LIB=device: the product library, or the tap build for the tools that drive a tap).  usage: bash tools/gpu_final.sh [tag]
set -u
TAG=${1:-final}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 ) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
export SGX_CAMPAIGN_LIB=device
c() { out=$1; lim=$2; shift 2; timeout $lim python "$@" 2>&1 | grep -v amdgpu.ids | tail -4 > $O/$out.txt; }
c orb 170 tools/campaign_orb.py 61 120 100000
c orb_geometry 170 tools/campaign_orb_geometry.py 62 120 100000
c match 150 tools/campaign_match.py 63 100 100000
c flow 150 tools/campaign_flow.py 64 100 100000
c tracker 170 tools/campaign_tracker.py 65 120 100000
c solvers 200 tools/campaign_solvers.py 66 150 100000
c ba_large 150 tools/campaign_ba_large.py 67 100 100000
c detection_output 150 tools/campaign_detection_output.py 68 100 100000
c detector_bf16x3 260 tools/campaign_detector.py 69 200
SGX_DET_GEMM=f32 c detector_f32 180 tools/campaign_detector.py 70 120
for f in $O/*.txt; do echo "$(basename $f): $(cat $f | tr '\n' ' ' | cut -c1-400)"; done
