"""diagnostic (round 6): two identical Python-orchestrated trackers + detectors side by side on the same frames; every intermediate buffer of the extraction stage is compared
after each step and a difference in the LK output is judged against the host entry run alone.  usage: [MODE=prio] [POLLUTE=64] python tools/diag_two_trackers.py [reps] [taps | lib.so]
MODE=prio restores the mixed stream priorities of rounds 2-5 (the condition under which the LK tracker differs, profiles/r6_lk_priority_diagnosis.md); MODE=keep / swap / addr: see the loop."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sg_slam_amd
from sg_slam_amd import synth
from sg_slam_amd.capi import DetResult
from sg_slam_amd.detector import Detector2D
from sg_slam_amd.tracker import TrackerBatch
from test_tracker_native_gpu import CAM
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pol = C.CDLL(os.path.join(ROOT, 'tools', 'lds_pollute', 'liblds_pollute.so')) if int(os.environ.get('POLLUTE', '0')) else None
from sg_slam_amd.capi import SgxLib
lib = (SgxLib(os.path.join(ROOT, 'tests', 'taps', 'libsgx_taps.so')) if sys.argv[2] == 'taps' else SgxLib(os.path.join(ROOT, sys.argv[2]))) if len(sys.argv) > 2 else sg_slam_amd.load()
param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
layers = synth.parse_ncnn_param(param); _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=-0.5)
S, MB, NF = 2, 100, 5
gen = synth.PlaneStream(seed=1234); offs = [3, 57]
frames = [[gen.frame(o + t) for o in offs] for t in range(NF)]
T0 = np.stack([gen.Tcw(o) for o in offs])
ABL = os.environ.get('ABL', '')      # ablations: nodet (no detector forward at all), anodet / bnodet (only for the first / second tracker), nolocal (no local-map stage), bextract (second tracker: no step, only its detector)
FIRST = [None]
class Side:
    def __init__(self):
        self.det = Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=S, lib=lib)
        self.tr = TrackerBatch(lib, S, CAM, xp='torch', lk=True, max_boxes=MB, local_map='nolocal' not in ABL); self.tr.set_initial_pose(T0)
        self.sD = torch.cuda.Stream()
        self.res = [torch.zeros((S, C.sizeof(DetResult)), dtype=torch.uint8, device='cuda') for _ in range(2)]
        self.boxes = [torch.zeros((S, MB, 4), dtype=torch.float32, device='cuda') for _ in range(2)]
        self.nb = [torch.zeros(S, dtype=torch.int32, device='cuda') for _ in range(2)]; self.have = [torch.zeros(S, dtype=torch.int32, device='cuda') for _ in range(2)]
        self.ev = [torch.cuda.Event() for _ in range(2)]
        self.dbg = None
        if hasattr(lib.dll, 'sgx_flow_debug_set_dbg'):
            self.dbg = torch.zeros((S, self.tr.cap, 4, 24), dtype=torch.float32, device='cuda')
            lib.dll.sgx_flow_debug_set_dbg(self.tr.flow.h, C.c_void_p(self.dbg.data_ptr()))
    def step(self, t, d_gray, d_depth, d_bgr):
        b = t & 1
        self.sD.wait_stream(torch.cuda.current_stream())
        if t >= 2: self.sD.wait_event(self.tr.ev_extract[(t - 2) % 3])
        if 'nodet' not in ABL and not ('bnodet' in ABL and self is not FIRST[0]) and not ('anodet' in ABL and self is FIRST[0]):
            self.det.detect_batch_dev(d_bgr, 640 * 3, S, self.res[b], self.boxes[b], self.nb[b], MB, self.have[b], stream=self.sD.cuda_stream)
        self.ev[b].record(self.sD)
        self.tr.step(d_gray, d_depth, mask=dict(boxes=self.boxes[b], nboxes=self.nb[b], have_dynamic=self.have[b], event=self.ev[b]))
    def snap(self, t):
        tr = self.tr; b = t & 1; c = tr.cur
        g = lambda x: x.cpu().numpy().copy()
        rn = g(tr.rn); d = dict(nb=g(self.nb[b]), boxes=g(self.boxes[b]), have=g(self.have[b]), rn=rn, n=g(tr.n[c]))
        for s in range(S):
            k = int(rn[s]); d['rkeys%d' % s] = g(tr.rkeys[s, :k]); d['prev_xy%d' % s] = g(tr.prev_xy[s, :k]).view(np.uint32); d['lk_status%d' % s] = g(tr.lk_status[s, :k]); d['keep%d' % s] = g(tr.keep[s, :k])
            d['keys%d' % s] = g(tr.keys[c][s, :int(d['n'][s])])
        d['F'] = g(tr.F).view(np.uint64); d['f_ok'] = g(tr.f_ok); d['f_stats'] = g(tr.f_stats); d['pre_boxes'] = g(tr.pre_boxes); d['pre_nboxes'] = g(tr.pre_nboxes); d['pre_have'] = g(tr.pre_have)
        d['Tcw'] = g(tr.Tcw[1]).view(np.uint32)
        return d
order = ['nb', 'boxes', 'have', 'rn', 'rkeys0', 'rkeys1', 'prev_xy0', 'prev_xy1', 'lk_status0', 'lk_status1', 'pre_nboxes', 'pre_boxes', 'pre_have', 'f_ok', 'F', 'f_stats', 'keep0', 'keep1', 'n', 'keys0', 'keys1', 'Tcw']
nbad = 0; bad_keys = []; keep_alive = []; MODE = os.environ.get('MODE', '')
if 'prio' in MODE: TrackerBatch.stream_priorities = (0, -1)      # the round 2-5 setting: tracking stream at high priority
for rep in range(reps):
    if 'swap' in MODE: B = Side(); A = Side()
    else: A, B = Side(), Side()
    held = []
    if 'keep' in MODE: keep_alive.append((A, B))
    if 'addr' in MODE: print('rep', rep, 'A prev_xy %x rkeys %x rn %x pyr %s | B prev_xy %x' % (A.tr.prev_xy.data_ptr(), A.tr.rkeys.data_ptr(), A.tr.rn.data_ptr(), '', B.tr.prev_xy.data_ptr()))
    for t in range(NF):
        fr = frames[t]
        d_gray = torch.from_numpy(np.stack([f[0] for f in fr])).cuda(); d_depth = torch.from_numpy(np.stack([f[1] for f in fr]).view(np.int16)).cuda()
        d_bgr = d_gray.unsqueeze(-1).expand(S, 480, 640, 3).contiguous(); held.append((d_gray, d_depth, d_bgr))
        if pol: assert pol.lds_pollute(C.c_uint32(0x7fc00000 + 977 * t + 31 * rep), int(os.environ['POLLUTE']), 400, 1024) == 0
        FIRST[0] = A
        A.step(t, d_gray, d_depth, d_bgr); B.step(t, d_gray, d_depth, d_bgr)
        torch.cuda.synchronize()
        if t == 0: continue
        a, b = A.snap(t), B.snap(t)
        diff = [k for k in order if a[k].shape != b[k].shape or not (a[k] == b[k]).all()]
        if diff:
            nbad += 1; k = diff[0]; print('rep %d t %d: differ %s; first: %s' % (rep, t, diff, k), flush=True)
            if a[k].shape == b[k].shape:
                w = np.argwhere(a[k] != b[k]); print('   where', w[:6].tolist(), 'A', a[k][tuple(w[0])], 'B', b[k][tuple(w[0])], 'count', len(w))
                if k.startswith('prev_xy'):
                    s = int(k[-1]); i = int(w[0][0]); print('   key', a['rkeys%d' % s][i].view(np.float32)[:2], 'A xy', a[k][i].view(np.float32), 'B xy', b[k][i].view(np.float32), 'status', a['lk_status%d' % s][i], b['lk_status%d' % s][i])
            if k.startswith('prev_xy'): bad_keys += np.argwhere((a[k] != b[k]).any(1)).ravel().tolist()
            if k.startswith('prev_xy'):      # which side is right: the host entry (fresh handle, alone) on the same frame pair and keypoints
                from sg_slam_amd.flow import OpticalFlowLK
                s_ = int(k[-1]); fl = OpticalFlowLK(width=640, height=480, max_batch=1, lib=lib)
                pts = a['rkeys%d' % s_].view(np.float32).reshape(len(a['rkeys%d' % s_]), 7)[:, :2].copy()
                truth, st_ = fl(frames[t][s_][0], frames[t - 1][s_][0], pts)
                tv = truth.view(np.uint32)
                wa = np.argwhere((a[k] != tv).any(1)).ravel().tolist(); wb = np.argwhere((b[k] != tv).any(1)).ravel().tolist()
                print('   against the host entry alone: A differs at', wa, ' B differs at', wb)
                for i_ in sorted(set(wa + wb))[:4]: print('      key %d truth %s A %s B %s' % (i_, truth[i_], a[k][i_].view(np.float32), b[k][i_].view(np.float32)))
            if k.startswith('prev_xy') and A.dbg is not None:
                s_ = int(k[-1]); da, db = A.dbg[s_].cpu().numpy(), B.dbg[s_].cpu().numpy()
                for i_ in np.argwhere((a[k] != b[k]).any(1)).ravel().tolist()[:3]:
                    for lv in (3, 2, 1, 0):
                        if not (da[i_, lv, :22].view(np.uint32) == db[i_, lv, :22].view(np.uint32)).all(): print('      key %d level %d A-matrix %s iters A %d B %d  longest gap between iterations (10 ns ticks, at j): A %d @%d  B %d @%d   median gap of all keys at this level: A %d B %d' % (i_, lv, 'same' if (da[i_, lv, :3] == db[i_, lv, :3]).all() else 'DIFF', da[i_, lv, 3], db[i_, lv, 3], da[i_, lv, 22], da[i_, lv, 23], db[i_, lv, 22], db[i_, lv, 23], np.median(da[:, lv, 22]), np.median(db[:, lv, 22]))); [print('         j %d  A b %s nextx %r | B b %s nextx %r %s   D %r %r' % (j_, da[i_, lv, 4 + 3 * j_:6 + 3 * j_].tolist(), float(da[i_, lv, 6 + 3 * j_]), db[i_, lv, 4 + 3 * j_:6 + 3 * j_].tolist(), float(db[i_, lv, 6 + 3 * j_]), '' if (da[i_, lv, 4 + 3 * j_:7 + 3 * j_].view(np.uint32) == db[i_, lv, 4 + 3 * j_:7 + 3 * j_].view(np.uint32)).all() else '<--', float(da[i_, lv, 21]), float(db[i_, lv, 21]))) for j_ in range(min(6, int(max(da[i_, lv, 3], db[i_, lv, 3], 1))))]
            if lib.has_taps:
                for slot in range(2):
                    for fr_ in range(S):
                        for lv in range(A.tr.flow.levels):
                            pa, pb = A.tr.flow.debug_level(slot, fr_, lv), B.tr.flow.debug_level(slot, fr_, lv)
                            if not (pa == pb).all(): print('   pyramid slot %d frame %d level %d differs at %d pixels, first %s' % (slot, fr_, lv, int((pa != pb).sum()), np.argwhere(pa != pb)[0].tolist()))
            break
print('reps', reps, 'bad', nbad, 'failing key indices', sorted(bad_keys))
