"""Differential campaign (CPU): random frame pairs / local maps, thresholds, ratio tests and observation patterns through both matchers of the kernel-logic
emulator and the oracle; match indices and in-view flags must be identical.  usage: python tools/campaign_match.py <seed> <seconds>
Round 1: 2 seeds x 1100 s + 2 x 3000 s = 8 133 cases, 0 mismatches."""
import sys, time; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from sg_slam_amd import synth
from sg_slam_amd.matcher import ORBmatcher
from sg_slam_amd.capi import SgxLib
from oracle import oracle as orc
from scenes import make_pair, make_local_map, CAM
from _campaign_lib import campaign_lib
lib, XP = campaign_lib()
seed0 = int(sys.argv[1]); rng = np.random.RandomState(seed0)
sf = orc.orb_params()['scale']
t0 = time.time(); n = 0; bad = 0
MAXC = int(sys.argv[3]) if len(sys.argv) > 3 else None          # optional: stop after this many cases (deterministic runs)
while time.time() - t0 < float(sys.argv[2]) and (MAXC is None or n < MAXC):
    S = synth.PlaneStream(seed=int(rng.randint(0, 100000)))
    t = int(rng.randint(0, 60))
    mode = ['zero', 'mixed', 'all'][rng.randint(0, 3)]
    cur, last = make_pair(orc, S, t, seed=int(rng.randint(0, 1000)), obs_mode=mode, pose_noise=float(rng.choice([0, 0.002, 0.01, 0.05])))
    th = float(rng.choice([7, 15, 30])); mono = bool(rng.rand() < 0.2); ori = bool(rng.rand() < 0.8)
    em, en = orc.search_by_projection_frame(cur, last, CAM, sf, th=th, mono=mono, check_ori=ori)
    c2 = dict(cur)
    gn = ORBmatcher(0.9, ori, lib=lib).SearchByProjection(c2, last, th, mono, CAM, sf)
    ok1 = gn == en and (c2['match'] == em).all()
    F, lm = make_local_map(orc, S, t + 2, seed=int(rng.randint(0, 1000)), n_prev=int(rng.randint(1, 4)))
    thl = float(rng.choice([3, 5])); nn = float(rng.choice([0.8, 0.6]))
    res = orc.search_by_projection_local(F, lm, CAM, sf, th=thl, nnratio=nn)
    F2 = dict(F); L2 = {k: v.copy() for k, v in lm.items()}
    gl = ORBmatcher(nn, True, lib=lib).SearchByProjectionLocal(F2, L2, thl, CAM, sf)
    ok2 = gl == res[1] and (F2['match_local'] == res[0]).all() and (L2['in_view'] == res[2]).all()
    n += 1
    if not (ok1 and ok2):
        bad += 1; print('MISMATCH', ok1, ok2, t, mode, th, mono, ori, thl, nn, flush=True)
print('seed', seed0, 'cases', n, 'bad', bad, flush=True)
