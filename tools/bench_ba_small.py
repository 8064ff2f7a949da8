import sys, os, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import sg_slam_amd
from sg_slam_amd.optimizer import Optimizer
from oracle import oracle as orc
from scenes import make_ba_problem, CAM
lib = sg_slam_amd.load()
prob, _, _ = make_ba_problem(orc, n_free=int(sys.argv[1]) if len(sys.argv) > 1 else 20, n_fixed=int(sys.argv[2]) if len(sys.argv) > 2 else 40, n_points=int(sys.argv[3]) if len(sys.argv) > 3 else 2000, seed=21)
def run():
    p2 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in prob.items()}
    t = time.perf_counter(); er, st = Optimizer.LocalBundleAdjustment(p2, CAM, lib=lib); return time.perf_counter() - t, st
run(); run()
ts = [run()[0] for _ in range(10)]
print(json.dumps(dict(ms_min=min(ts) * 1e3, ms_med=sorted(ts)[5] * 1e3)))
