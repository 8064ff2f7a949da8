#!/bin/bash
# round 6, GPU session F: LK after the fp64 detours were removed: bit-exactness tests, smoke, stand-alone timing, VALU count
set -u
O=gpurun_out/r6f; mkdir -p $O
( timeout 600 python -m pytest tests/test_flow_gpu.py tests/test_tracker_native_gpu.py -q -p no:cacheprovider -x 2>&1 | tail -5 ) > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python tools/bench_flow.py --streams 512 2>/dev/null | tail -3 | tee $O/flow.txt
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -o p -- python $GRAFT_REPO_ROOT/tools/bench_flow.py --streams 512 --reps 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/pmc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -A5 "k_lk_trackN" | tee $O/pmc_lk.txt
find $O -name "*.csv" -size +2M -delete
