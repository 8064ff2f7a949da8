#!/bin/bash
# device-side differential campaigns after the round's kernel changes (blur tile walk, bf16x3 pointwise layers): a few minutes each
set -u
TAG=${1:-camp}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
export SGX_CAMPAIGN_LIB=device
timeout 200 python tools/campaign_orb.py 41 90 100000 2>&1 | tail -1 > $O/orb.txt
timeout 200 python tools/campaign_orb_geometry.py 42 90 100000 2>&1 | tail -1 > $O/orb_geometry.txt
timeout 400 python tools/campaign_detector.py 43 240 2>&1 | tail -4 > $O/detector_bf16x3.txt
SGX_DET_GEMM=f32 timeout 300 python tools/campaign_detector.py 44 120 2>&1 | tail -4 > $O/detector_f32.txt
timeout 200 python tools/campaign_tracker.py 45 90 100000 2>&1 | tail -1 > $O/tracker.txt
timeout 200 python tools/campaign_flow.py 46 60 100000 2>&1 | tail -1 > $O/flow.txt
for f in $O/*.txt; do echo "$(basename $f): $(cat $f | tr '\n' ' ' | cut -c1-400)"; done
