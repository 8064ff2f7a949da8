"""Build profiles/<name>_traffic.json from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE and WRITE_SIZE, separate passes as the
MI355X guide prescribes).  Per bench kernel class: mean HBM-side bytes per bench launch = sum over the class's dispatches / launches, where
FETCH_SIZE (KB) is doubled (gfx950 reports half of a wide coalesced read stream) and WRITE_SIZE (KB) is taken as is.
usage: pmc_traffic.py fetch.csv write.csv frames_per_launch steps_profiled out.json"""
import csv, json, sys, collections
CLASS = [('k_pyramid', 'pyramid_resize'), ('k_resize', 'pyramid_resize'), ('k_fast_cells', 'fast_cells'), ('k_octree', 'octree'), ('k_orient_desc', 'orient_desc'),
         ('k_stereo_from_rgbd', 'stereo_from_rgbd'), ('k_motion_model', 'motion_model'), ('k_match_project_frame', 'match_project_frame'),
         ('k_match_project_local', 'match_project_local'), ('k_pose_opt', 'pose_opt'), ('k_unproject', 'unproject'),
         ('k_make_map_points', 'map_point_glue'), ('k_merge_matches', 'map_point_glue'), ('k_gather_xw', 'map_point_glue'),
         ('k_dynamic_mask', 'dynamic_mask'), ('k_compact_keys', 'dynamic_mask')]
# bench launches per step of each class (a "launch" in bench.py's per_kernel table = one sgx_* call)
PER_STEP = dict(pyramid_resize=1, fast_cells=1, octree=1, orient_desc=1, stereo_from_rgbd=1, motion_model=1, match_project_frame=1, match_project_local=1,
                pose_opt=2, unproject=1, map_point_glue=3, dynamic_mask=2)
def total(path, counter):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        for pre, cls in CLASS:
            if r['Kernel_Name'].split('(')[0].split('<')[0].strip().endswith(pre) or r['Kernel_Name'].startswith(pre) or (' ' + pre) in r['Kernel_Name']:
                acc[cls] += float(r['Counter_Value']); break
    return acc
fetch, write, S, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
f = total(fetch, 'FETCH_SIZE'); w = total(write, 'WRITE_SIZE')
res = {}
for cls in sorted(set(f) | set(w)):
    launches = PER_STEP[cls] * steps
    res[cls] = int(round((2.0 * f.get(cls, 0.0) + w.get(cls, 0.0)) * 1024.0 / launches))
json.dump({'note': 'HBM-side bytes per bench launch from rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes over bench.py (separate passes); FETCH_SIZE (KB) doubled per the '
                   'gfx950 correction in MI355X_MICROARCH.md, WRITE_SIZE (KB) as reported; all steps of the profiled run (warm-up included) are averaged',
           'frames_per_launch': S, 'steps_profiled': steps, 'bytes_per_launch': res}, open(out, 'w'), indent=1)
print(json.dumps(res))
