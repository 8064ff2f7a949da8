"""diagnostic (round 6): per-keypoint, per-level records of the LK tracker (debug build sg_slam_amd/ab/libsgx_lkdbg.so: gradient matrix, 1 / det, b and position of the first six
iterations) from a quiet run against a run beside tools/lds_pollute's k_corun (EXT_KIND: instruction classes): which quantity goes wrong first?
The debug build's record macro (SGX_LK_DBG) lived in the diagnostic tree of commit d376699 and is not in the sources any more; the output that mattered is kept as
profiles/r6_lk_dbg_values.txt (per-lane partial sums differ in the last 16-lane row only: how the packed-fp32 cause was found)."""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sg_slam_amd import synth
from sg_slam_amd.capi import SgxLib
from sg_slam_amd.flow import OpticalFlowLK
from sg_slam_amd.orb import ORBextractor
lib = SgxLib(os.path.join(ROOT, 'sg_slam_amd', 'ab', 'libsgx_lkdbg.so')); ext = C.CDLL(os.path.join(ROOT, 'tools', 'lds_pollute', 'liblds_pollute.so'))
kind = int(os.environ.get('EXT_KIND', '2'))
S = 2; gen = synth.PlaneStream(seed=1234); offs = [3, 57]
f0 = torch.from_numpy(np.stack([gen.frame(o + 1)[0] for o in offs])).cuda(); f1 = torch.from_numpy(np.stack([gen.frame(o + 2)[0] for o in offs])).cuda()
ex = ORBextractor(nfeatures=1000, width=640, height=480, max_batch=S, lib=lib); cap = ex.capacity
keys = torch.zeros((S, cap, 28), dtype=torch.uint8, device='cuda'); desc = torch.zeros((S, cap, 32), dtype=torch.uint8, device='cuda'); n = torch.zeros(S, dtype=torch.int32, device='cuda')
ex.extract_batch_dev(f1, 640, S, keys, desc, n); torch.cuda.synchronize(); nn = n.cpu().numpy()
fl = OpticalFlowLK(width=640, height=480, max_batch=S, lib=lib); st = torch.cuda.Stream()
xy = torch.zeros((S, cap, 2), dtype=torch.float32, device='cuda'); status = torch.zeros((S, cap), dtype=torch.uint8, device='cuda')
dbgall = torch.zeros(S * cap * 4 * 24 + S * cap * 4 * 64, dtype=torch.float32, device='cuda'); dbg = dbgall[:S * cap * 4 * 24].view(S, cap, 4, 24); lanes = dbgall[S * cap * 4 * 24:].view(torch.int32).view(S, cap, 4, 32, 2); lib.dll.sgx_flow_debug_set_dbg(fl.h, C.c_void_p(dbg.data_ptr()))
LAST = [None]
def run(co):
    dbgall.zero_(); torch.cuda.synchronize()
    if co: assert ext.corun_launch(1024, 2000, kind, 8, 3) == 0
    fl.reset(); fl.lk_batch_dev(f0, 640, S, None, None, cap, None, None, stream=st.cuda_stream); fl.lk_batch_dev(f1, 640, S, keys, n, cap, xy, status, stream=st.cuda_stream)
    torch.cuda.synchronize(); LAST[0] = lanes.cpu().numpy().copy(); return dbg.cpu().numpy().copy(), xy.cpu().numpy().copy()
d0, x0 = run(False); l0 = LAST[0]; d0b, x0b = run(False)
print('quiet vs quiet: records identical', (d0[..., :22].view(np.uint32) == d0b[..., :22].view(np.uint32)).all())
names = ['A11', 'A12', 'A22', 'iters'] + sum([['b1_%d' % j, 'b2_%d' % j, 'x_%d' % j] for j in range(6)], []) + ['invdet']
first = {}; shown = []; vals = []
for rep in range(5):
    d1, x1 = run(True); l1 = LAST[0]
    for s in range(S):
        for k in range(int(nn[s])):
            if (x1[s, k].view(np.uint32) == x0[s, k].view(np.uint32)).all(): continue
            done = False
            for lv in (3, 2, 1, 0):
                a, b = d0[s, k, lv, :22].view(np.uint32), d1[s, k, lv, :22].view(np.uint32)
                order = [0, 1, 2, 21] + [4 + 3 * j + i for j in range(6) for i in (2, 0, 1)]      # A, invdet, then per iteration: position, b1, b2
                for f in order:
                    if a[f] != b[f]:
                        nm = names[f] if f < 22 else '?'; key = ('level %d' % lv, nm.split('_')[0], 'odd key' if k & 1 else 'even key')
                        if len(vals) < 24 and nm.startswith('b1'):
                            j_ = int(nm.split('_')[1]); q_, c_ = d0[s, k, lv], d1[s, k, lv]
                            vals.append('key %d level %d iteration %d: b1 quiet %r (%s) contaminated %r (%s) | b2 quiet %r contaminated %r | b * 2^20: %r -> %r, %r -> %r' % (k, lv, j_, float(q_[4 + 3 * j_]), q_[4 + 3 * j_:5 + 3 * j_].view(np.uint32)[0].item().to_bytes(4, 'big').hex(), float(c_[4 + 3 * j_]), c_[4 + 3 * j_:5 + 3 * j_].view(np.uint32)[0].item().to_bytes(4, 'big').hex(), float(q_[5 + 3 * j_]), float(c_[5 + 3 * j_]), float(q_[4 + 3 * j_]) * 2 ** 20, float(c_[4 + 3 * j_]) * 2 ** 20, float(q_[5 + 3 * j_]) * 2 ** 20, float(c_[5 + 3 * j_]) * 2 ** 20))
                        first[key] = first.get(key, 0) + 1; done = True
                        if nm == 'b1_0' or nm == 'b2_0':      # first iteration of a level: the per-lane partial sums are on record
                            dl = np.argwhere((l0[s, k, lv] != l1[s, k, lv]).any(1)).ravel().tolist()
                            kk = ('per-lane partial sums of that iteration', 'all equal (the reduction differs)' if not dl else 'differ in %d lanes' % len(dl)); first[kk] = first.get(kk, 0) + 1
                            if dl and len(shown) < 6: shown.append((s, k, lv, dl[:8], l0[s, k, lv][dl[:4]].tolist(), l1[s, k, lv][dl[:4]].tolist()))
                        break
                if done: break
print('first differing quantity (top level first) of the keypoints whose result differs, 5 contaminated runs:')
for k_, v in sorted(first.items(), key=lambda t: -t[1]): print('  ', k_, v)
for t in shown: print('   stream %d key %d level %d: lanes of the half-wave %s quiet %s contaminated %s' % t)
for v in vals: print('  ', v)
