#!/bin/bash
# round 6, GPU session C: tile / occupancy alternatives of k_hrb (tap build, SGX_HRB_PICK) on the per-step harness
set -u
O=gpurun_out/r6c; mkdir -p $O
for p in ${PICKS:-0 1 2}; do SGX_HRB_PICK=$p timeout 300 python tools/prof_det_ops.py 512 5 2>/dev/null | head -${ROWS:-9} > $O/pick$p.txt; echo "== pick $p"; cat $O/pick$p.txt; done
SGX_DET_HRB=0 timeout 300 python tools/prof_det_ops.py 512 5 2>/dev/null | head -15 > $O/fb2.txt; echo "== k_fused_block2 / per-layer"; cat $O/fb2.txt
