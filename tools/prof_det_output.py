"""Detector2D::detect alone on the bench's frames: standalone time of the forward graph and of DetectionOutput (+ candidate statistics).
usage: python tools/prof_det_output.py [batch] [reps]      (under rocprofv3 --kernel-trace --stats for per-kernel durations)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import sg_slam_amd
from sg_slam_amd import synth
from sg_slam_amd.detector import Detector2D
from sg_slam_amd.capi import DetResult
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
PARAM = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
lib = sg_slam_amd.load()
layers = synth.parse_ncnn_param(PARAM); _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=-4.0)
det = Detector2D(0.9, 0.01, param_text=open(PARAM).read(), bin_bytes=blob, max_batch=B, lib=lib)
gen = synth.LayeredStream(seed=1234)
frames = np.stack([gen.frame(7 * s)[0] for s in range(min(B, 16))])
gray = torch.from_numpy(frames).cuda()[torch.arange(B) % len(frames)]
bgr = gray.unsqueeze(-1).expand(B, 480, 640, 3).contiguous()
res = torch.zeros((B, C.sizeof(DetResult)), dtype=torch.uint8, device='cuda')
boxes = torch.zeros((B, 8, 4), dtype=torch.float32, device='cuda'); nb = torch.zeros(B, dtype=torch.int32, device='cuda'); have = torch.zeros(B, dtype=torch.int32, device='cuda')
lib.dll.sgx_profile_enable(1)
side = torch.cuda.Stream() if os.environ.get('SGX_TOOL_STREAM') else None          # a non-default stream takes the captured-graph path of the forward
for r in range(REPS + 1):
    det.detect_batch_dev(bgr, 640 * 3, B, res, boxes, nb, 8, have, stream=None if side is None else side.cuda_stream)
    torch.cuda.synchronize()
    if r == 0:
        ms = np.zeros(lib.dll.sgx_profile_num_classes(), 'f4'); n = np.zeros(len(ms), 'i4'); lib.dll.sgx_profile_read(ms.ctypes.data, n.ctypes.data, 1)
ms = np.zeros(lib.dll.sgx_profile_num_classes(), 'f4'); n = np.zeros(len(ms), 'i4'); lib.dll.sgx_profile_read(ms.ctypes.data, n.ctypes.data, 1)
lib.dll.sgx_profile_class_name.restype = C.c_char_p
for k in range(len(ms)):
    if n[k]: print('%-14s %.4f ms per launch (%d launches, batch %d)' % (lib.dll.sgx_profile_class_name(k).decode(), ms[k] / n[k], n[k], B))
R = np.frombuffer(res.cpu().numpy().tobytes(), dtype=np.uint8).reshape(B, -1)
nraw = R[:, :4].copy().view('i4')[:, 0]
print('n_raw per frame: mean %.1f min %d max %d; person boxes mean %.2f' % (nraw.mean(), nraw.min(), nraw.max(), nb.float().mean().item()))
