"""Micro-benchmark of sgx_match_project_frame_batch_dev (64 identical frame pairs) under a few parameter variations."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import sg_slam_amd
from sg_slam_amd import synth
from sg_slam_amd.capi import _vp, KP_DTYPE
from sg_slam_amd.matcher import camera_struct
from oracle import oracle as orc
from scenes import make_pair, CAM
from _campaign_lib import tool_lib; lib = tool_lib()
S = synth.PlaneStream(seed=1234)
cur, last = make_pair(orc, S, 3, seed=1, obs_mode='zero')
B, cap = 64, 1024
def rep(a, dtype=None, shape_tail=()):
    out = np.zeros((B, cap) + shape_tail, a.dtype if dtype is None else dtype)
    out[:, :len(a)] = a
    return out
dev = lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda()
ck = dev(rep(cur['keys'])); cd = dev(rep(cur['desc'], shape_tail=(32,))); cu = dev(rep(cur['uright'])); cn = dev(np.full(B, len(cur['keys']), 'i4'))
lk = dev(rep(last['keys'])); ln = dev(np.full(B, len(last['keys']), 'i4')); lh = dev(rep(last['has_mp'])); lo = dev(rep(np.zeros(len(last['keys']), np.uint8)))
lx = dev(rep(last['xw'], shape_tail=(3,))); lb = dev(rep(last['obs'])); lm = dev(rep(last['mpdesc'], shape_tail=(32,)))
cT = dev(np.tile(cur['Tcw'].reshape(1, 16), (B, 1)).astype('f4')); lT = dev(np.tile(last['Tcw'].reshape(1, 16), (B, 1)).astype('f4'))
match = torch.zeros((B, cap), dtype=torch.int32, device='cuda'); nm = torch.zeros(B, dtype=torch.int32, device='cuda')
sf = np.ascontiguousarray(orc.orb_params()['scale'], 'f4'); cs = camera_struct(CAM)
def run(th, ori, label):
    f = lambda: lib.check(lib.dll.sgx_match_project_frame_batch_dev(B, cap, _vp(ck), _vp(cd), _vp(cu), _vp(cn), _vp(cT), _vp(lk), _vp(ln), _vp(lh), _vp(lo), _vp(lx), _vp(lb), _vp(lm), _vp(lT),
                                                                     C.byref(cs), _vp(sf), 8, float(th), 0, ori, _vp(match), _vp(nm), None))
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print(label, 'th', th, 'ori', ori, '%.1f us' % ((time.perf_counter() - t) / 20 * 1e6), 'matches', int(nm[0]))
run(15, 1, 'normal'); run(0.001, 1, 'tiny window'); run(15, 0, 'no ori'); run(30, 1, 'wide')
lh.zero_(); run(15, 1, 'no map points')
