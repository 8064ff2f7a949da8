#!/bin/bash
# round 6, GPU session M: micro-reproducer of the mixed-stream-priority effect outside the library (tools/ubench/prio_lanes.hip)
set -u
O=gpurun_out/r6m; mkdir -p $O
for m in 5 1 2 3 4; do for p in 1 0; do timeout 120 tools/ubench/prio_lanes $m 300 $p 1024 2>&1 | tail -1 | tee -a $O/prio_lanes.txt; done; done
