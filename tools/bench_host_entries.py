"""Latency of the host-pointer (one frame, synchronous) entry points a reference maintainer binds first: ORBextractor::operator(), SearchByProjection,
PoseOptimization through the Python mirror over the C ABI (ms per call, median of 30).  usage: python tools/bench_host_entries.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import sg_slam_amd
from sg_slam_amd import synth
from sg_slam_amd.orb import ORBextractor
from sg_slam_amd.matcher import ORBmatcher
from sg_slam_amd.optimizer import Optimizer
from oracle import oracle as orc
from scenes import make_pair, make_pose_problem, CAM

lib = sg_slam_amd.load()
S = synth.PlaneStream(seed=1234)
g, _, _ = S.frame(3)
ex = ORBextractor(lib=lib)
cur, last = make_pair(orc, S, 5, seed=1, obs_mode='zero')
sf = orc.orb_params()['scale']; is2 = orc.orb_params()['inv_sigma2']
fr, _, _ = make_pose_problem(orc, n=800, seed=44)
m = ORBmatcher(0.9, True, lib=lib)

def med(f, n=30):
    for _ in range(3): f()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))

out = dict(orb_extract_ms=med(lambda: ex(g)), search_by_projection_ms=med(lambda: m.SearchByProjection(dict(cur), last, 15, False, CAM, sf)),
           pose_optimization_ms=med(lambda: Optimizer.PoseOptimization({k: (v.copy() if hasattr(v, 'copy') else v) for k, v in fr.items()}, CAM, is2, lib=lib)))
print(json.dumps(out))
