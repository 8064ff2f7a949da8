#!/bin/bash
# round 6, GPU session M: every detector plan step as the only co-runner of the LK tracker (tap build, 20 launches of the step beside 4 LK runs, 60 repetitions each)
set -u
O=gpurun_out/r6m; mkdir -p $O; : > $O/steps.txt
for i in $(seq 0 51); do
  r=$(LKRUNS=4 CORUN=step STEP=$i STEP_REPS=${STEP_REPS:-20} STEP_BATCH=${STEP_BATCH:-2} timeout 120 python tools/diag_lk_repeat.py 60 taps 2>&1 | grep -v amdgpu.ids | grep -E "^co-runner|^reps" | tr '\n' ' ')
  echo "step $i: $r" | cut -c1-200 | tee -a $O/steps.txt
done
