cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o b -- python $R/bench.py --no-cpu-baseline --no-config2 --no-config4 --no-host-input --no-detector --no-pipeline --steps 16 --warmup 2 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/ks/**/b_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('k_blur_levels', 'k_orient_desc4', 'k_fast_cells', 'k_pyramid', 'k_octree<true, 256, 2048>')):
        print(n[:28], r['Calls'], round(float(r['AverageNs'])/1e6, 4), 'ms')
PY
