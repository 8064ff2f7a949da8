"""Differential campaign (CPU): LocalBundleAdjustment with 23 - 90 free keyframes (138 - 540 unknowns: the blocked Cholesky; SGX_TUNE_CHOL_WIDE_MIN=0 also sends them
through the two-level / wide-update path), 1 500 - 5 000 landmarks, through the kernel-logic emulator and the oracle: identical LM iteration counts and erase flags, poses
within 1e-5, landmarks by tests/test_localba.py::points_close.  usage: [SGX_TUNE_CHOL_WIDE_MIN=0] python tools/campaign_ba_large.py <seed> <seconds>
Round 1 (3 + 3 seeds x 700 s, 1 x 3000 s): 9 950 problems, 12 reports — each a keyframe left with 2 - 4 edges after the outlier pass (its pose is not determined by the data): final chi2
equal to 4e-7, every other keyframe within 1e-5."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from sg_slam_amd.capi import SgxLib
from sg_slam_amd.optimizer import Optimizer
from oracle import oracle as orc
from scenes import make_ba_problem, CAM
from test_localba import close, points_close
from _campaign_lib import campaign_lib
lib, XP = campaign_lib()
rng = np.random.RandomState(int(sys.argv[1])); t0 = time.time(); n = bad = 0
MAXC = int(sys.argv[3]) if len(sys.argv) > 3 else None          # optional: stop after this many cases (deterministic runs)
while time.time() - t0 < float(sys.argv[2]) and (MAXC is None or n < MAXC):
    nfree = int(rng.choice([23, 30, 44, 50, 64, 90])); nfix = int(rng.choice([3, 10, 25])); npts = int(rng.choice([1500, 3000, 5000]))
    seed = int(rng.randint(0, 1 << 30)); bo = float(rng.choice([0.0, 0.08, 0.2])); bm = float(rng.choice([0.0, 0.2, 0.6]))
    prob, _, _ = make_ba_problem(orc, n_free=nfree, n_fixed=nfix, n_points=npts, seed=seed, outlier_frac=bo, mono_frac=bm)
    ep, ex_, ee, et, ei = orc.local_ba(prob, CAM)
    p2 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in prob.items()}
    er, st = Optimizer.LocalBundleAdjustment(p2, CAM, lib=lib)
    ok = st['iterations'] == tuple(ei) and (er == ee).all() and close(p2['poses'], ep) and points_close(p2['points'], ex_)
    n += 1
    if not ok: bad += 1; print('MISMATCH seed', seed, nfree, nfix, npts, bo, bm, st['iterations'], tuple(int(v) for v in ei), int((er != ee).sum()), close(p2['poses'], ep), points_close(p2['points'], ex_), flush=True)
print('seed', sys.argv[1], 'cases', n, 'bad', bad, flush=True)
