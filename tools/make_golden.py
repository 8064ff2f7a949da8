"""Regenerates tests/golden/*.npz: outputs of the CPU oracle on fixed synthetic inputs.  They are NOT reference outputs (the reference cannot be
built here) — they freeze the oracle so that an accidental change of the parity yardstick itself is caught (tests/test_golden.py).
usage: python tools/make_golden.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle as orc
from sg_slam_amd import synth
from scenes import make_pair, make_pose_problem, make_big_ba_problem, CAM

out = os.path.join(ROOT, 'tests', 'golden')
S = synth.PlaneStream(seed=1234)
g, d, T = S.frame(5)
k, desc = orc.orb_extract(g)
np.savez_compressed(os.path.join(out, 'orb_frame5.npz'), keys=k, desc=desc, gray_crc=np.array([int(np.bitwise_xor.reduce(g.astype(np.uint64).ravel() * np.arange(1, g.size + 1, dtype=np.uint64)))], np.uint64))
cur, last = make_pair(orc, S, 9, 8)
m, n = orc.search_by_projection_frame(cur, last, CAM, orc.orb_params()['scale'], th=15)
np.savez_compressed(os.path.join(out, 'match_frames_9_8.npz'), match=m, n=np.array([n]))
fr, _, _ = make_pose_problem(orc, n=400, seed=43)
en, eT, eout = orc.pose_optimization(fr, CAM, orc.orb_params()['inv_sigma2'])
np.savez_compressed(os.path.join(out, 'poseopt_n400_seed43.npz'), n=np.array([en]), T=eT, outlier=eout)
# BASELINE config 4 (2 000 keyframes / 50 000 landmarks): the oracle's LocalBundleAdjustment on the seeded generator of tests/scenes.py — possible since the oracle's reduced
# system is solved in its envelope (seconds; the dense LDL^T of rounds 1-3 needed hours at 11 994 unknowns).  Kept: LM iteration counts, the LM trace (chi2, lambda, trials per
# iteration), the erase flags (bit-packed), every pose, every 16th point.  The inputs are regenerated from the seed by the tests.
prob, _, _ = make_big_ba_problem(2000, 50000)
bposes, bpts, berase, btrace, biters = orc.local_ba(prob, CAM)
np.savez_compressed(os.path.join(out, 'ba_2000kf_50klm.npz'), iters=biters, trace=btrace, erase_bits=np.packbits(berase), n_edges=np.array([len(berase)]), poses=bposes.astype('f4'),
                    points_every16=bpts[::16].astype('f4'), edge_pose_crc=np.array([int(np.bitwise_xor.reduce(prob['edge_pose'].astype(np.uint64) * np.arange(1, len(berase) + 1, dtype=np.uint64)))], np.uint64))
print('written', sorted(os.listdir(out)))
