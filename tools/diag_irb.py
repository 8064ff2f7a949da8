"""Diagnostic (GPU): the detector plan with the matrix-core inverted-residual block kernels (k_irb) against the per-layer plan, blob by blob.
usage: python tools/diag_irb.py [batch]   -> prints every blob present in both plans with its bit-exactness / max abs difference"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sg_slam_amd
from sg_slam_amd.detector import Detector2D
from sg_slam_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PARAM = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
from _campaign_lib import taps_lib
lib = taps_lib()          # the tap build (include/sgx_debug.h): plan selection / per-step timing / blob read-back are not in the product library
layers = synth.parse_ncnn_param(PARAM); W, blob = synth.synth_ncnn_weights(layers)
rng = np.random.RandomState(5)
imgs = rng.randint(0, 256, (B, 480, 640, 3)).astype(np.uint8)
names = []
for L in layers:
    for o in L['outs'] if 'outs' in L else L.get('outputs', []):
        names.append(o)
outs = {}
for irb in (False, True):
    det = Detector2D(0.9, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=B, lib=lib, irb=irb)
    det.detect_batch(imgs)
    outs[irb] = {n: np.stack([det.debug_blob(n, b) for b in range(B)]) for n in names if det.has_blob(n)}
    print('irb', irb, 'kernels', det.num_kernels, 'blobs', len(outs[irb]))
    det.close()
bad = 0
for n in names:
    if n in outs[False] and n in outs[True]:
        a, b = outs[False][n], outs[True][n]
        same = (a.view(np.uint32) == b.view(np.uint32)).all()
        if not same:
            bad += 1
            d = np.abs(a.astype(np.float64) - b.astype(np.float64))
            nz = np.argwhere(d > 0)
            print(f'{n:>20s} DIFF max {d.max():.3e} (ref max {np.abs(a).max():.3e}) count {len(nz)}/{a.size} first {nz[0].tolist()} eq-as-float {(a == b).all()} nan {np.isnan(b).sum()}')
print('blobs differing:', bad)
