import sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.getcwd())
import sg_slam_amd
from sg_slam_amd import synth, dist as sdist
from sg_slam_amd.tracker import TrackerBatch
from sg_slam_amd.detector import Detector2D
from sg_slam_amd.capi import DetResult
lib = sg_slam_amd.load(); cam = dict(synth.TUM3)
S, T, MB = 256, 6, 100
gen = synth.LayeredStream(seed=1234); t0s = sdist.stream_offsets(0, S)
order = list(range(T)) + list(range(T - 2, 0, -1))
host, hdep = synth.synth_streams('LayeredStream', 1234, t0s, T)
d_frames = torch.from_numpy(host).cuda(); d_depth_all = torch.from_numpy(hdep.view(np.int16)).cuda()
param = 'tests/golden/mobilenetv3_ssdlite_voc.param'
layers = synth.parse_ncnn_param(param); _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)
det = Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=S, lib=lib)
d_bgr = d_frames.unsqueeze(-1).expand(T, S, 480, 640, 3).contiguous()
tr = TrackerBatch(lib, S, cam, xp='torch', lk=True, max_boxes=MB)
tr.set_initial_pose(np.stack([gen.Tcw(t0) for t0 in t0s]))
res = torch.zeros((S, C.sizeof(DetResult)), dtype=torch.uint8, device='cuda'); boxes = torch.zeros((S, MB, 4), dtype=torch.float32, device='cuda')
nb = torch.zeros(S, dtype=torch.int32, device='cuda'); have = torch.zeros(S, dtype=torch.int32, device='cuda')
hist = []
for i in range(28):
    fi = order[i % len(order)]
    det.detect_batch_dev(d_bgr[fi], 640 * 3, S, res, boxes, nb, MB, have, stream=None)
    torch.cuda.synchronize()
    tr.step(d_frames[fi], d_depth_all[fi], mask=dict(boxes=boxes, nboxes=nb, have_dynamic=have))
    tr.synchronize()
    n, nm, ninl = tr.last_counts(); nml, ninl2 = tr.last_local_counts()
    hist.append((tr.rn.cpu().numpy().copy(), n.copy(), nm.copy(), ninl.copy(), ninl2.copy(), nb.cpu().numpy().copy(), tr.f_stats.cpu().numpy().copy(), tr.f_ok.cpu().numpy().copy()))
bad = [s for s in range(S) if any(h[4][s] < 30 for h in hist[1:])]
print('bad streams', bad)
for s in bad[:3]:
    for i, h in enumerate(hist):
        print(s, i, 'raw', h[0][s], 'kept', h[1][s], 'match', h[2][s], 'inl1', h[3][s], 'inl2', h[4][s], 'boxes', h[5][s], 'ransac', h[6][s], 'ok', h[7][s])
print('mean kept', np.mean([h[1].mean() for h in hist[1:]]), 'mean boxes', np.mean([h[5].mean() for h in hist]))
err = []
P = tr.last_pose()
gt = np.stack([gen.Tcw(t0 + order[27 % len(order)]) for t0 in t0s])
e = np.abs(P - gt).reshape(S, -1).max(1); print('pose err: median', np.median(e), 'max', e.max(), 'n>0.05', (e > 0.05).sum())
