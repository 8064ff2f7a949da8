"""diagnostic (round 6): the native tracker + detector against the Python orchestration, repeated; prints which quantity diverges first.  usage: python tools/diag_native_det.py [reps] [taps]"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sg_slam_amd
from sg_slam_amd import synth
from sg_slam_amd.capi import DetResult, SgxLib
from sg_slam_amd.detector import Detector2D
from sg_slam_amd.tracker import TrackerBatch
from sg_slam_amd.tracker_native import TrackerNative
from test_tracker_native_gpu import CAM
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pol = C.CDLL(os.path.join(ROOT, 'tools', 'lds_pollute', 'liblds_pollute.so')) if os.environ.get('POLLUTE') else None
lib = (SgxLib(os.path.join(ROOT, 'tests', 'taps', 'libsgx_taps.so')) if sys.argv[2] == 'taps' else SgxLib(os.path.join(ROOT, sys.argv[2]))) if len(sys.argv) > 2 else sg_slam_amd.load()
param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
layers = synth.parse_ncnn_param(param); _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=-0.5)
def mkdet(S): return Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=S, lib=lib)
S, MB, NF = 2, 100, 5
gen = synth.PlaneStream(seed=1234); offs = [3, 57]
frames = [[gen.frame(o + t) for o in offs] for t in range(NF)]
T0 = np.stack([gen.Tcw(o) for o in offs])
ref = {}
for rep in range(reps):
    det_py, det_nat = mkdet(S), mkdet(S)
    py = TrackerBatch(lib, S, CAM, xp='torch', lk=True, max_boxes=MB); py.set_initial_pose(T0)
    nat = TrackerNative(lib, S, CAM, dynamic_mask=True, max_boxes=MB, detector=det_nat); nat.set_initial_pose(T0)
    sD = torch.cuda.Stream()
    res = [torch.zeros((S, C.sizeof(DetResult)), dtype=torch.uint8, device='cuda') for _ in range(2)]
    boxes = [torch.zeros((S, MB, 4), dtype=torch.float32, device='cuda') for _ in range(2)]
    nb = [torch.zeros(S, dtype=torch.int32, device='cuda') for _ in range(2)]; have = [torch.zeros(S, dtype=torch.int32, device='cuda') for _ in range(2)]
    ev = [torch.cuda.Event() for _ in range(2)]
    held = []; msg = []
    for t in range(NF):
        fr = frames[t]
        d_gray = torch.from_numpy(np.stack([f[0] for f in fr])).cuda(); d_depth = torch.from_numpy(np.stack([f[1] for f in fr]).view(np.int16)).cuda()
        d_bgr = d_gray.unsqueeze(-1).expand(S, 480, 640, 3).contiguous()
        held.append((d_gray, d_depth, d_bgr))
        b = t & 1
        if pol and rep > 0: assert pol.lds_pollute(C.c_uint32(0x7fc00000 + 977 * t + rep), int(os.environ['POLLUTE']), 400, 1024) == 0
        sD.wait_stream(torch.cuda.current_stream())
        if t >= 2: sD.wait_event(py.ev_extract[(t - 2) % 3])
        det_py.detect_batch_dev(d_bgr, 640 * 3, S, res[b], boxes[b], nb[b], MB, have[b], stream=sD.cuda_stream)
        ev[b].record(sD)
        py.step(d_gray, d_depth, mask=dict(boxes=boxes[b], nboxes=nb[b], have_dynamic=have[b], event=ev[b]))
        nat.step(d_gray, d_depth, d_bgr=d_bgr, stream=torch.cuda.current_stream().cuda_stream)
        bxs, bns = [], []
        torch.cuda.current_stream().synchronize()
        for s in range(S):
            bx = torch.zeros((MB, 4), dtype=torch.float32, device='cuda'); bn = torch.zeros(1, dtype=torch.int32, device='cuda'); torch.cuda.current_stream().synchronize()
            nat.snapshot_boxes(s, bx, bn); bxs.append(bx); bns.append(bn)
        r = nat.read(); py.synchronize(); sD.synchronize(); torch.cuda.synchronize()
        n, nm, ninl = py.last_counts()
        nbn = np.array([int(x.item()) for x in bns]); nbp = nb[b].cpu().numpy()
        same_boxes = all((bxs[s].cpu().numpy() == boxes[b][s].cpu().numpy()).all() for s in range(S))
        line = 't%d nb nat %s py %s boxes %s | nkeys nat %s py %s' % (t, nbn, nbp, 'same' if same_boxes else 'DIFF', r['nkeys'], n)
        if t > 0: line += ' | raw %s %s f_stats %s' % (r['nkeys_raw'], py.rn.cpu().numpy(), 'same' if (r['f_stats'] == py.f_stats.cpu().numpy()).all() else 'DIFF nat %s py %s' % (r['f_stats'].tolist(), py.f_stats.cpu().numpy().tolist()))
        ok = (r['nkeys'] == n).all() and same_boxes and (nbn == nbp).all()
        cur = dict(nat=(r['nkeys'].tolist(), r['f_stats'].tolist() if t > 0 else None), py=(n.tolist(), py.f_stats.cpu().numpy().tolist() if t > 0 else None))
        if rep == 0: ref[t] = cur
        else:
            for k in ('nat', 'py'):
                if cur[k] != ref[t][k] or ref[t]['nat'] != ref[t]['py']: line += ' | %s deviates from the quiet run: %s vs %s' % (k, cur[k], ref[t][k]); ok = False
        msg.append(('   ' if ok else '!! ') + line)
    bad = any(m.startswith('!!') for m in msg)
    print('rep', rep, 'BAD' if bad else 'ok', flush=True)
    if bad: print('\n'.join(msg), flush=True)
    nat.close()
