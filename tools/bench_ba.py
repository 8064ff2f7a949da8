"""Times sgx_local_bundle_adjustment on LocalBA-sized synthetic graphs (SURVEY §8(d) input 4) next to the oracle on one host core.
Writes one JSON line (profiles/)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sg_slam_amd
from sg_slam_amd.optimizer import Optimizer
from oracle import oracle as orc
from scenes import make_ba_problem, CAM

lib = sg_slam_amd.load()
out = []
for (nf, nx, npt) in ((20, 40, 2000), (60, 60, 6000), (120, 80, 12000)):
    prob, _, _ = make_ba_problem(orc, n_free=nf, n_fixed=nx, n_points=npt, seed=21)
    ne = len(prob['edge_pose'])
    def run_gpu():
        p2 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in prob.items()}
        t = time.perf_counter(); er, st = Optimizer.LocalBundleAdjustment(p2, CAM, lib=lib); return time.perf_counter() - t, st
    run_gpu()
    tg, st = min((run_gpu() for _ in range(3)), key=lambda r: r[0])
    t = time.perf_counter(); orc.local_ba(prob, CAM); tc = time.perf_counter() - t
    its = sum(st['iterations'])
    out.append(dict(free_kf=nf, fixed_kf=nx, points=npt, edges=ne, lm_iterations=its, gpu_ms=tg * 1e3, cpu_oracle_ms_1core=tc * 1e3,
                    edges_per_s=ne * its / tg, alg_bytes_per_edge=192, achieved_GBs=192 * ne * its / tg / 1e9))
print(json.dumps(dict(bench='local_bundle_adjustment', results=out)))
