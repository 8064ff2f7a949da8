#!/bin/bash
# last GPU trip of a round: the -m gpu suite on the final tree (log kept under profiles/), then bounded device campaigns
set -u
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -2 $O/gpu_tests.log
export SGX_CAMPAIGN_LIB=device
timeout 100 python tools/campaign_orb.py 51 60 100000 2>&1 | tail -1 > $O/orb.txt
timeout 150 python tools/campaign_detector.py 53 110 2>&1 | tail -4 > $O/detector_bf16x3.txt
timeout 100 python tools/campaign_tracker.py 55 60 100000 2>&1 | tail -1 > $O/tracker.txt
timeout 100 python tools/campaign_flow.py 56 50 100000 2>&1 | tail -1 > $O/flow.txt
for f in $O/*.txt; do echo "$(basename $f): $(cat $f | tr '\n' ' ' | cut -c1-400)"; done
