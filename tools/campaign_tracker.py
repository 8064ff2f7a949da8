"""Differential campaign (CPU): the batched tracking harness (ORB -> stereo -> motion model -> SearchByProjection -> PoseOptimization -> local-map search ->
PoseOptimization -> unproject, frame after frame, two streams) on random synthetic streams and start offsets, stage by stage against the chained oracle
(tests/test_tracker_emu.py::run_tracker raises on the first difference).  usage: python tools/campaign_tracker.py <seed> <seconds>
Round 1 (6 seeds x 700 s + 3 seeds x 2400 s): more than 4 000 runs of 4 - 7 frames on two streams each, 0 differences."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from sg_slam_amd.capi import SgxLib
from oracle import oracle as orc
from test_tracker_emu import run_tracker
from _campaign_lib import campaign_lib
lib, XP = campaign_lib()
rng = np.random.RandomState(int(sys.argv[1])); t0 = time.time(); n = bad = 0
MAXC = int(sys.argv[3]) if len(sys.argv) > 3 else None          # optional: stop after this many cases (deterministic runs)
while time.time() - t0 < float(sys.argv[2]) and (MAXC is None or n < MAXC):
    ss = int(rng.randint(0, 100000)); offs = (int(rng.randint(0, 80)), int(rng.randint(0, 80))); nf = int(rng.randint(4, 8))
    try:
        run_tracker(lib, orc, XP, stream_seed=ss, offs=offs, nframes=nf)
    except AssertionError as e:
        bad += 1; print('MISMATCH stream_seed', ss, 'offs', offs, 'frames', nf, repr(e)[:200], flush=True)
    n += 1
print('seed', sys.argv[1], 'runs', n, 'bad', bad, flush=True)
