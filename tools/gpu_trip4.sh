#!/bin/bash
# blur traffic experiment: persistent walk (grid 4096) against one tile per workgroup (the round-2 order), standalone time + HBM fetch of k_blur_levels, and the full bench both ways
set -u
TAG=${1:-trip4}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
PB="python $R/bench.py --streams 512 --no-cpu-baseline --no-config2 --no-config4 --no-host-input --no-detector --steps 3 --warmup 1"
cd /tmp; export TMPDIR=/tmp
for g in 4096 1000000 ${EXTRA_GRIDS:-}; do
  SGX_TUNE_ORB_BLUR_GRID=$g timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$g -o p -- $PB > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find $O/fetch_$g -name "*counter_collection.csv" | head -1) 2>/dev/null | grep -A1 -E "k_blur_levels|k_orient_desc4|k_fast_cells|k_pyramid" > $O/fetch_$g.txt
  rm -rf $O/fetch_$g
  cd $R
  SGX_TUNE_ORB_BLUR_GRID=$g timeout 300 python bench.py --no-cpu-baseline --no-detector --no-config2 --no-config4 --no-host-input --no-pipeline --steps 48 --warmup 4 > $O/serial_$g.json 2>/dev/null
  SGX_TUNE_ORB_BLUR_GRID=$g timeout 300 python bench.py --no-cpu-baseline --no-config2 --no-config4 --no-host-input --steps 60 --warmup 6 > $O/full_$g.json 2>/dev/null
  cd /tmp
done
cd $R
python - <<PY
import json
for g in "4096 1000000 ${EXTRA_GRIDS:-}".split():
    try:
        s = json.load(open("$O/serial_%s.json" % g)); f = json.load(open("$O/full_%s.json" % g))
        print('grid', g, 'orient_desc standalone', s['roofline']['per_kernel']['orient_desc']['avg_ms_per_launch'], 'in pipeline', f['roofline']['per_kernel']['orient_desc']['avg_ms_per_launch'], 'full', round(f['value']), 'fps', round(f['ms_per_step'], 3), 'ms')
    except Exception as e: print(g, 'failed', e)
    print(open("$O/fetch_%s.txt" % g).read())
PY
