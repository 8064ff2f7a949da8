"""Build profiles/<name>_traffic.json from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and WRITE_SIZE, separate passes as the
MI355X guide prescribes).  Per bench kernel class: mean HBM-side bytes per bench launch = sum over the class's dispatches / launches, where
FETCH_SIZE (KB) is doubled (gfx950 reports half of a wide coalesced read stream) and WRITE_SIZE (KB) is taken as is.
usage: pmc_traffic.py [--det det_fetch.csv det_write.csv launches_per_step] fetch.csv write.csv frames_per_launch steps_profiled out.json [bench.json: launches per step per class are taken from its per_kernel table]"""
import csv, json, sys, collections
from pmc_classes import classify, steps_ran, base_name
# bench launches per step of each class (a "launch" in bench.py's per_kernel table = one sgx_* call / one prof begin-end bracket)
PER_STEP = dict(pyramid_resize=1, fast_cells=1, octree=1, orient_desc=1, stereo_from_rgbd=1, motion_model=1, match_project_frame=1, match_project_local=1,
                pose_opt=2, unproject=1, map_point_glue=3, dynamic_mask=2, lk_pyramid=1, lk_track=1, fm_ransac=1, det_forward=1, det_output=1)
RAN = {}
def total(path, counter):
    acc = collections.defaultdict(float); disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        cls = classify(r['Kernel_Name'])
        if cls is not None: acc[cls] += float(r['Counter_Value']); disp[cls][base_name(r['Kernel_Name'])].add(r['Dispatch_Id'])
    for cls, d in disp.items(): RAN[cls] = steps_ran({k: len(v) for k, v in d.items()})
    return acc
det = None
if '--det' in sys.argv:
    i = sys.argv.index('--det'); det = (sys.argv[i + 1], sys.argv[i + 2], int(sys.argv[i + 3])); del sys.argv[i:i + 4]
fetch, write, S, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
if len(sys.argv) > 6:
    bj = json.load(open(sys.argv[6]))
    for k, e in bj['roofline']['per_kernel'].items(): PER_STEP[k] = max(1, int(round(e['launches'] / bj['steps'])))
f = total(fetch, 'FETCH_SIZE'); w = total(write, 'WRITE_SIZE')
res = {}
for cls in sorted(set(f) | set(w)):
    launches = PER_STEP[cls] * min(steps, RAN.get(cls, steps))          # the tracking-stage classes do not run in the first step of a run
    res[cls] = int(round((2.0 * f.get(cls, 0.0) + w.get(cls, 0.0)) * 1024.0 / launches))
if det:            # detector classes from the per-step profile (tools/prof_det_ops.py: every plan step launched det[2] times on its own)
    fd = total(det[0], 'FETCH_SIZE'); wd = total(det[1], 'WRITE_SIZE')
    for cls in ('det_forward', 'det_output'):
        if cls in fd or cls in wd: res[cls] = int(round((2.0 * fd.get(cls, 0.0) + wd.get(cls, 0.0)) * 1024.0 / det[2]))
json.dump({'note': 'HBM-side bytes per bench launch from rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes over bench.py (separate passes); FETCH_SIZE (KB) doubled per the '
                   'gfx950 correction in MI355X_MICROARCH.md, WRITE_SIZE (KB) as reported; all steps of the profiled run (warm-up included) are averaged',
           'frames_per_launch': S, 'steps_profiled': steps, 'bytes_per_launch': res}, open(out, 'w'), indent=1)
print(json.dumps(res))
