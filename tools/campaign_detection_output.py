"""Differential campaign (CPU): ncnn DetectionOutput + detect() filtering alone (k_det_class_nms, k_det_merge through sgx_det_debug_detection_output) on random head
outputs — box offsets of three spreads, score distributions with heavy ties, sparse and dense candidate sets, empty classes — through the kernel-logic emulator against
the oracle: identical rows, labels, scores and order; boxes within 1e-5.  usage: python tools/campaign_detection_output.py <seed> <seconds>
Round 1: 2 seeds x 900 s + 1 x 3000 s = 7 067 cases, 0 mismatches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import detector_oracle as D
from sg_slam_amd.detector import Detector2D
from sg_slam_amd.capi import SgxLib, DetResult
from _campaign_lib import campaign_lib
lib, XP = campaign_lib(taps=True)
PARAM = ROOT + '/tests/golden/mobilenetv3_ssdlite_voc.param'
layers = D.parse_param(PARAM); W, blob = D.synth_weights(layers, seed=7)
det = Detector2D(0.90, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=1, lib=lib)
n, nc = det.num_priors, det.num_class
rng0 = np.random.RandomState(0)
img = rng0.randint(0, 256, (480, 640, 3)).astype(np.uint8)
_, blobs = D.forward(layers, W, D.preprocess(img)); priors = blobs['mbox_priorbox']
p = [L for L in layers if L['type'] == 'DetectionOutput'][0]['p']
seed0 = int(sys.argv[1]); rng = np.random.RandomState(seed0); t0 = time.time(); cases = bad = 0
MAXC = int(sys.argv[3]) if len(sys.argv) > 3 else None          # optional: stop after this many cases (deterministic runs)
while time.time() - t0 < float(sys.argv[2]) and (MAXC is None or cases < MAXC):
    loc = (rng.randn(1, n, 4) * rng.choice([0.1, 0.5, 2.0])).astype('f4')
    raw = (rng.rand(n, nc) ** rng.choice([1, 3, 8])).astype('f4')
    if rng.rand() < 0.5: raw = np.round(raw * 16) / np.float32(16)                       # heavy score ties
    raw[rng.rand(n) < rng.choice([0.0, 0.5, 0.95, 0.999])] *= 0.001
    for c in rng.choice(nc, rng.randint(0, 6), replace=False): raw[:, c] = 0
    conf = np.ascontiguousarray(raw[None], 'f4')
    res = (DetResult * 1)()
    lib.check(lib.tap('sgx_det_debug_detection_output')(det.h, loc.ctypes.data, conf.ctypes.data, 1, res), 'do')
    exp = D.detection_output(loc[0].reshape(-1), conf[0].reshape(-1), priors, p)
    r = res[0]
    got = np.array([[d.label, d.score, d.xmin, d.ymin, d.xmax, d.ymax] for d in r.raw[:r.n_raw]], np.float32).reshape(-1, 6)
    ok = got.shape == exp.shape and (len(exp) == 0 or ((got[:, :2] == exp[:, :2]).all() and np.abs(got[:, 2:] - exp[:, 2:]).max() < 1e-5))
    cases += 1
    if not ok: bad += 1; print('MISMATCH', cases, got.shape, exp.shape, flush=True)
print('seed', seed0, 'cases', cases, 'bad', bad, flush=True)
