"""Times the batched detector forward (sgx_det_forward_batch_dev) on synthetic 640x480x3 frames; reports frames/s and
achieved TFLOP/s on the MFMA-eligible pointwise convolutions.  One JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ctypes as C
import sg_slam_amd
from sg_slam_amd.detector import Detector2D
from sg_slam_amd.capi import _vp
from oracle import detector_oracle as D          # only to synthesise the weight blob (the reference's .bin is absent)

PARAM = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
if os.environ.get('SGX_BENCH_TAPS_LIB') == '1':          # A/B runs: the tap build honours the SGX_* switches (SGX_DET_FORK, SGX_DET_GEMM, ...)
    sys.path.insert(0, os.path.join(ROOT, 'tools')); from _campaign_lib import taps_lib; lib = taps_lib()
else:
    lib = sg_slam_amd.load()
layers = D.parse_param(PARAM); W, blob = D.synth_weights(layers)
out = []
BATCHES = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1, 16, 64]
for B in BATCHES:
    det = Detector2D(0.9, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=B, lib=lib)
    img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device='cuda')
    ts = torch.cuda.Stream(); st = ts.cuda_stream          # a non-default stream: the plan is replayed as a hipGraph
    ts.wait_stream(torch.cuda.current_stream())
    dl = C.c_void_p(); dc = C.c_void_p()
    def fwd(): lib.check(lib.dll.sgx_det_forward_batch_dev(det.h, _vp(img), 640 * 3, B, C.byref(dl), C.byref(dc), _vp(st)))
    for _ in range(3): fwd()
    torch.cuda.synchronize(); t = time.perf_counter(); n = 20
    for _ in range(n): fwd()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    out.append(dict(batch=B, ms_per_forward=dt * 1e3, frames_per_s=B / dt, kernels_per_forward=det.num_kernels, gflop_per_frame=2 * det.gmac,
                    achieved_TFLOPs=2 * det.gmac * B / dt / 1e3, fp32_mfma_peak_TFLOPs=157.3))
    det.close()
print(json.dumps(dict(bench='detector_forward', results=out)))
