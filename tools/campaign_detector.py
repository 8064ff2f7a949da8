"""Differential campaign (CPU, ~6 s per case): the detector (fused or one-kernel-per-layer plan) with RANDOM weight draws (each calibrated like the harness's: He draw + synthetic
batch-norm fold, sg_slam_amd.synth.synth_ncnn_weights) and random images through the kernel-logic emulator — or, SGX_CAMPAIGN_LIB=device, the tap build on the GPU — against the numpy
oracle, by the criterion of tests/test_detector.py::run_compare: early blobs to 1e-5; every tapped blob within max(4 x the oracle's own fp32 drift, 6e-6) of a float64 run; DetectionOutput
exact on the device's own head outputs and, end to end, equal to the oracle's rows up to ties within 2e-5 / knife-edge IoUs.   usage: python tools/campaign_detector.py <seed> <seconds>
Round 5 (calibrated draws): the first criterion tried (2 x, floor 2e-6) reported 10 of 95 cases, all with ratios 2.0 - 2.8 at the heads (an ascending-k fp32 chain against numpy's blocked
sums) — then 3 x / 4e-6 reported 3 of 325 (ratios 3.2 - 4.2 at the 128-value blob '944'; one row list whose scores differ by 1e-5) — the source of 4 x / 6e-6 and the 2e-5 row tolerance;
totals in profiles/r5_campaigns.md.  Rounds 1-4 (raw He draws, chaotic networks): profiles/HISTORY_r1-r4.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import detector_oracle as D
from sg_slam_amd.capi import SgxLib
import test_detector as T
from _campaign_lib import campaign_lib
lib, XP = campaign_lib(taps=True)
layers = D.parse_param(T.PARAM)
seed0 = int(sys.argv[1]); t0 = time.time(); n = bad = 0
rng = np.random.RandomState(seed0)
while time.time() - t0 < float(sys.argv[2]):
    W, blob = D.synth_weights(layers, seed=int(rng.randint(0, 100000)))
    s = int(rng.randint(10, 100000))
    try:
        T.run_compare(lib, (layers, W, blob), seeds=(s,), fuse=bool(rng.rand() < 0.7))
    except AssertionError as e:
        bad += 1; print('MISMATCH image seed', s, repr(e)[:300], flush=True)
    n += 1
print('seed', seed0, 'cases', n, 'bad', bad, flush=True)
