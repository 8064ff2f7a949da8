#!/bin/bash
# Re-collects the measurement artefacts kept under profiles/ on a 1-GPU MI355X box (run from the repo root; ~4 minutes of GPU time):
#   tools/collect_profiles.sh <tag>        e.g.  /usr/local/graft/bin/gpurun --timeout 1500 -- 'tools/collect_profiles.sh r2_a'
# Writes into gpurun_out/<tag>/; copy what should be judged into profiles/ afterwards.  Counter passes follow MI355X_MICROARCH.md: one counter per pass
# (FETCH_SIZE and WRITE_SIZE together made rocprofv3 abort on this pool), --kernel-trace only (no sys / runtime trace domains), each under a hard timeout.
set -u
TAG=${1:-run}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
# 1. the default bench line (with the CPU baseline) and the smaller / larger stream counts
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python bench.py --no-cpu-baseline --streams 64  > $O/bench_s64.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --streams 512 > $O/bench_s512.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --detector   > $O/bench_with_detector.json 2>/dev/null
# 2. kernel statistics of the same command (rocprofv3 wants a writable cwd / TMPDIR)
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
# 3. HBM traffic: separate passes, 8 steps each (2 warm-up + 6 timed), joined per bench kernel class by tools/pmc_traffic.py
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $O/fetch/p_counter_collection.csv $O/write/p_counter_collection.csv 256 8 $O/traffic.json
# 4. bundle adjustment: LocalBA sizes, 500 and 2000 keyframes, kernel statistics of the big one
timeout 100 python tools/bench_ba.py > $O/localba.json 2>/dev/null
timeout 100 python tools/bench_ba_big.py 500 12000 > $O/ba_500.json 2>/dev/null
timeout 100 python tools/bench_ba_big.py > $O/ba_2000.json 2>/dev/null
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ba_stats -o b -- python $R/tools/bench_ba_big.py > /dev/null 2>&1
cd $R
# 5. detector forward alone (frames/s per batch size, per-step table)
timeout 200 python tools/bench_det.py > $O/detector_bench.json 2>/dev/null
python - <<PY
import json
j = json.load(open("$O/bench_default.json"))
print("default bench:", round(j["value"]), "frames/s,", round(j["ms_per_step"], 3), "ms/step; dominant kernel", j["roofline"]["kernel"], "frac", j["roofline"]["frac"])
PY
