"""Diagnostic (GPU): the detector with bf16x3 matrix products (k_conv_pw3 / k_irb3, sgx_det_bf16.h) against the exact-fp32 plan, blob by blob.
usage: python tools/diag_gemm.py [batch]   -> for the per-layer plan, the default plan and the all-shapes k_irb plan: every blob both runs hold with its max
difference relative to the blob's magnitude; anything above 1e-4 is a defect (a layout error shows as O(1)), 1e-6 .. 1e-5 is summation-order noise."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sg_slam_amd
from sg_slam_amd.detector import Detector2D
from sg_slam_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
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
worst_all = 0.0
for plan, kw in (('per-layer', dict(fuse=False, irb=False)), ('fused, no irb', dict(fuse=True, irb=False)), ('default', dict(fuse=True, irb=None)), ('irb everywhere', dict(fuse=True, irb=True))):
    outs = {}
    for gemm in ('f32', 'bf16x3'):
        det = Detector2D(0.9, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=B, lib=lib, gemm=gemm, **kw)
        det.detect_batch(imgs)
        outs[gemm] = {n: np.stack([det.debug_blob(n, b) for b in range(B)]) for n in names if det.has_blob(n)}
        nk = det.num_kernels
        det.close()
    print(f'== plan {plan}: {nk} kernels, {len(outs["f32"])} blobs')
    bad = 0
    for n in names:
        if n in outs['f32'] and n in outs['bf16x3']:
            a, b = outs['f32'][n].astype(np.float64), outs['bf16x3'][n].astype(np.float64)
            d = np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)
            worst_all = max(worst_all, d if np.isfinite(d) else 1e9)
            flag = 'BAD' if (not np.isfinite(d) or d > 1e-4) else ('' if d > 0 else 'same')
            if flag == 'BAD': bad += 1
            print(f'{n:>22s} rel {d:.3e} (max |x| {np.abs(a).max():.3e}) nan {int(np.isnan(b).sum())} {flag}')
    print(f'== plan {plan}: {bad} blobs above 1e-4')
print('worst relative difference over all plans:', worst_all)
