"""BASELINE config 4 (2 000 keyframes / 50 000 landmarks): seconds per sgx_local_bundle_adjustment call with the dense and the envelope solver, plus the per-class
kernel time (sgx_profile).  usage: python tools/bench_ba_phases.py [n_kf n_points]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sg_slam_amd
from sg_slam_amd.optimizer import Optimizer
from scenes import CAM, make_big_ba_problem
NKF = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
NPT = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
from _campaign_lib import taps_lib
lib = taps_lib()          # the tap build (include/sgx_debug.h): plan selection / per-step timing / blob read-back are not in the product library
t = time.perf_counter(); prob, Ts, poses0 = make_big_ba_problem(NKF, NPT); gen_s = time.perf_counter() - t
out = dict(bench='bundle_adjustment_phases', keyframes=NKF, landmarks=NPT, edges=int(len(prob['edge_pose'])), generator_seconds=gen_s, solvers={})
for name, mode in (('dense', 1), ('auto', 0)):
    lib.tap('sgx_ba_debug_set_solver')(mode)
    for rep in range(2):
        p = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in prob.items()}
        if rep == 1: lib.profile_read(reset=True); lib.profile_enable(True)
        t = time.perf_counter(); er, st = Optimizer.LocalBundleAdjustment(p, CAM, lib=lib); dt = time.perf_counter() - t
    lib.profile_enable(False); prof = lib.profile_read()
    kern = {k: dict(ms=round(v[0], 3), groups=int(v[1])) for k, v in prof.items() if k.startswith('ba_') and v[1]}
    out['solvers'][name] = dict(seconds=dt, lm_iterations=[int(x) for x in st['iterations']], chi2=[float(x) for x in st['chi2']], kernel_ms=kern,
                                kernel_ms_total=round(sum(v['ms'] for v in kern.values()), 3), erased=int(np.asarray(er).sum()),
                                max_abs_translation_error=float(np.abs(p['poses'].astype('f8')[:, :3, 3] - Ts[:, :3, 3]).max()))
lib.tap('sgx_ba_debug_set_solver')(-1)
print(json.dumps(out))
