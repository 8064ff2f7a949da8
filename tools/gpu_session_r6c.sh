#!/bin/bash
# round 6, GPU session C: tile / occupancy alternatives of k_hrb (tap build, SGX_HRB_PICK) on the per-step harness
set -u
O=gpurun_out/r6c; mkdir -p $O
for p in 0 1 2; do SGX_HRB_PICK=$p timeout 300 python tools/prof_det_ops.py 512 5 2>/dev/null | head -7 | tail -5 > $O/pick$p.txt; echo "== pick $p"; cat $O/pick$p.txt; done
