"""diagnostic (round 6): the LK tracker on the SAME frame pair and keypoints over and over, beside a second instance on another stream; any change of the output between repetitions
is a race inside the kernel (or its pyramid).  usage: python tools/diag_lk_repeat.py [reps] [taps]      env: SGX_LK_KPW (tap build), POLLUTE=KB"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import sg_slam_amd
from sg_slam_amd import synth
from sg_slam_amd.capi import SgxLib
from sg_slam_amd.flow import OpticalFlowLK
from sg_slam_amd.orb import ORBextractor
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = (SgxLib(os.path.join(ROOT, 'tests', 'taps', 'libsgx_taps.so')) if sys.argv[2] == 'taps' else SgxLib(os.path.join(ROOT, sys.argv[2]))) if len(sys.argv) > 2 else sg_slam_amd.load()
pol = C.CDLL(os.path.join(ROOT, 'tools', 'lds_pollute', 'liblds_pollute.so')) if int(os.environ.get('POLLUTE', '0')) else None
S = 2; gen = synth.PlaneStream(seed=1234); offs = [3, 57]
f0 = torch.from_numpy(np.stack([gen.frame(o + 1)[0] for o in offs])).cuda(); f1 = torch.from_numpy(np.stack([gen.frame(o + 2)[0] for o in offs])).cuda()
ex = ORBextractor(nfeatures=1000, width=640, height=480, max_batch=S, lib=lib); cap = ex.capacity
keys = torch.zeros((S, cap, 28), dtype=torch.uint8, device='cuda'); desc = torch.zeros((S, cap, 32), dtype=torch.uint8, device='cuda'); n = torch.zeros(S, dtype=torch.int32, device='cuda')
ex.extract_batch_dev(f1, 640, S, keys, desc, n); torch.cuda.synchronize()
nn = n.cpu().numpy(); print('keys', nn)
class Inst:
    def __init__(self):
        self.fl = OpticalFlowLK(width=640, height=480, max_batch=S, lib=lib); self.st = torch.cuda.Stream()
        self.xy = torch.zeros((S, cap, 2), dtype=torch.float32, device='cuda'); self.status = torch.zeros((S, cap), dtype=torch.uint8, device='cuda')
    def run(self):
        self.fl.reset()
        self.fl.lk_batch_dev(f0, 640, S, None, None, cap, None, None, stream=self.st.cuda_stream)
        self.fl.lk_batch_dev(f1, 640, S, keys, n, cap, self.xy, self.status, stream=self.st.cuda_stream)
    def out(self):
        return [np.concatenate([self.xy[s, :nn[s]].cpu().numpy().view(np.uint32), self.status[s, :nn[s], None].cpu().numpy().astype(np.uint32)], 1) for s in range(S)]
A, B = Inst(), Inst()
corun = os.environ.get('CORUN', 'lk')
ext = C.CDLL(os.path.join(ROOT, 'tools', 'lds_pollute', 'liblds_pollute.so')) if 'ext' in corun else None      # CORUN=ext: tools/lds_pollute's k_corun (EXT_KIND bits: instruction classes of k_hrb)
if 'step' in corun:      # one detector plan step, launched STEP_REPS times on a side stream beside every LK run (tap build): which kernel is the one LK must not meet?
    from sg_slam_amd.detector import Detector2D
    param_ = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
    layers_ = synth.parse_ncnn_param(param_); _, blob_ = synth.synth_ncnn_weights(layers_, seed=7, person_logit=-0.5)
    SB = int(os.environ.get('STEP_BATCH', '2'))
    dets = Detector2D(0.9, 0.01, param_text=open(param_).read(), bin_bytes=blob_, max_batch=SB, lib=lib); sStep = torch.cuda.Stream()
    bgrs = torch.randint(0, 256, (SB, 480, 640, 3), dtype=torch.uint8, device='cuda')
    dets.time_ops(bgrs, SB, reps=1)      # every blob of the plan holds data
    STEP = int(os.environ['STEP']); STEP_REPS = int(os.environ.get('STEP_REPS', '20'))
    print('co-runner:', dets.op_descriptions()[STEP])
if 'prio' in corun:
    sH = torch.cuda.Stream(priority=-1); evA = torch.cuda.Event(); exH = ORBextractor(nfeatures=1000, width=640, height=480, max_batch=S, lib=lib)
    kH = torch.zeros_like(keys); dH = torch.zeros_like(desc); nH = torch.zeros_like(n)
if 'det' in corun:
    from sg_slam_amd.detector import Detector2D
    from sg_slam_amd.capi import DetResult
    param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
    layers = synth.parse_ncnn_param(param); _, blob = synth.synth_ncnn_weights(layers, seed=7, person_logit=-0.5)
    det = Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=S, lib=lib); sDet = torch.cuda.Stream()
    bgr = f1.unsqueeze(-1).expand(S, 480, 640, 3).contiguous()
    dres = torch.zeros((S, C.sizeof(DetResult)), dtype=torch.uint8, device='cuda'); dbox = torch.zeros((S, 100, 4), dtype=torch.float32, device='cuda')
    dnb = torch.zeros(S, dtype=torch.int32, device='cuda'); dhave = torch.zeros(S, dtype=torch.int32, device='cuda')
    if 'det2' in corun:
        det2 = Detector2D(0.9, 0.01, param_text=open(param).read(), bin_bytes=blob, max_batch=S, lib=lib); sDet2 = torch.cuda.Stream()
        dres2 = torch.zeros_like(dres); dbox2 = torch.zeros_like(dbox); dnb2 = torch.zeros_like(dnb); dhave2 = torch.zeros_like(dhave)
if 'orb' in corun:
    ex2 = ORBextractor(nfeatures=1000, width=640, height=480, max_batch=S, lib=lib); sOrb = torch.cuda.Stream()
    k2 = torch.zeros_like(keys); d2 = torch.zeros_like(desc); n2 = torch.zeros_like(n)
A.run(); torch.cuda.synchronize(); ref = A.out()
bad = 0
for r in range(reps):
    if pol: pol.lds_pollute(C.c_uint32(0x7fc00000 + r), int(os.environ['POLLUTE']), 50, 1024)
    if 'det' in corun: det.detect_batch_dev(bgr, 640 * 3, S, dres, dbox, dnb, 100, dhave, stream=sDet.cuda_stream)
    if 'det2' in corun and 'late' not in corun: det2.detect_batch_dev(bgr, 640 * 3, S, dres2, dbox2, dnb2, 100, dhave2, stream=sDet2.cuda_stream)
    if 'orb' in corun and 'first' in corun: ex2.extract_batch_dev(f1, 640, S, k2, d2, n2, stream=sOrb.cuda_stream)
    nrun = int(os.environ.get('LKRUNS', '1'))
    if 'ext' in corun: assert ext.corun_launch(int(os.environ.get('EXT_BLOCKS', '1024')), int(os.environ.get('EXT_ITERS', '2000')), int(os.environ['EXT_KIND']), int(os.environ.get('EXT_LDS_KB', '8')), int(os.environ.get('EXT_LAUNCHES', '3'))) == 0
    if 'step' in corun: lib.check(lib.tap('sgx_det_debug_run_step')(dets.h, C.c_void_p(bgrs.data_ptr()), 640 * 3, SB, STEP, STEP_REPS, C.c_void_p(sStep.cuda_stream)))
    outs = []
    for q_ in range(nrun):
        A.run()
        if nrun > 1:
            with torch.cuda.stream(A.st): outs.append((A.xy.clone(), A.status.clone()))
    if 'orb' in corun and 'first' not in corun: ex2.extract_batch_dev(f1, 640, S, k2, d2, n2, stream=sOrb.cuda_stream)
    if 'det2' in corun and 'late' in corun: det2.detect_batch_dev(bgr, 640 * 3, S, dres2, dbox2, dnb2, 100, dhave2, stream=sDet2.cuda_stream)
    if 'lk' in corun: B.run()
    if 'prio' in corun:      # a high-priority stream that waits for A's LK and then works (the tracking stream of rounds 2-5)
        evA.record(A.st); sH.wait_event(evA); exH.extract_batch_dev(f1, 640, S, kH, dH, nH, stream=sH.cuda_stream)
    torch.cuda.synchronize()
    cands = [('A', A)] + ([('B', B)] if 'lk' in corun else [])
    for q_, (xy_, st_) in enumerate(outs[:-1]):
        I_ = Inst.__new__(Inst); I_.xy = xy_; I_.status = st_; cands.append(('A#%d' % q_, I_))
    for name, I in cands:
        o = I.out()
        for s in range(S):
            w = np.argwhere((o[s] != ref[s]).any(1))
            if len(w):
                bad += 1; i = int(w[0][0]); k = keys[s, i].cpu().numpy().view(np.float32)
                print('rep %d %s stream %d: %d keys differ; key %d at (%.2f, %.2f) octave %d: %s vs quiet %s' % (r, name, s, len(w), i, k[0], k[1], keys[s, i].cpu().numpy().view(np.int32)[5], o[s][i, :2].view(np.float32), ref[s][i, :2].view(np.float32)), flush=True)
print('reps', reps, 'bad', bad)
