"""diagnostic (round 6): WHERE does the LK difference beside a dense bf16 matrix-product co-runner (tools/lds_pollute k_corun, kind 2) come from?
  MASK=none|disjoint|same   CU masks of the LK stream and the co-runner's stream (disjoint: the two never share a CU; same: both confined to one half of the chip)
  VICTIM=lk|orb             lk: pyramid + tracker (tap build: the pyramids of the contaminated run are read back and compared with the quiet run's as well); orb: the ORB extractor
  LIB=path                  a tap build other than tests/taps/libsgx_taps.so (tools/ab_build_taps.sh)
  EXT_KIND / EXT_BLOCKS / EXT_ITERS / EXT_LAUNCHES: the co-runner
usage: python tools/diag_lk_where.py [reps]"""
import os, sys, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sg_slam_amd import synth
from sg_slam_amd.capi import SgxLib
from sg_slam_amd.flow import OpticalFlowLK
from sg_slam_amd.orb import ORBextractor
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = SgxLib(os.path.join(ROOT, os.environ.get('LIB', 'tests/taps/libsgx_taps.so'))); ext = C.CDLL(os.path.join(ROOT, 'tools', 'lds_pollute', 'liblds_pollute.so'))
ext.corun_make_stream.restype = C.c_void_p; ext.corun_make_stream.argtypes = [C.c_int]; ext.corun_set_stream.argtypes = [C.c_void_p]
mask = os.environ.get('MASK', 'none'); victim = os.environ.get('VICTIM', 'lk')
kind = int(os.environ.get('EXT_KIND', '2')); blocks = int(os.environ.get('EXT_BLOCKS', '1024')); iters = int(os.environ.get('EXT_ITERS', '2000')); launches = int(os.environ.get('EXT_LAUNCHES', '3'))
sV = ext.corun_make_stream({'none': -1, 'disjoint': 0, 'same': 0}[mask]); sC = ext.corun_make_stream({'none': -1, 'disjoint': 1, 'same': 0}[mask])
assert sV and sC, 'stream creation failed'
ext.corun_set_stream(sC)
S = 2; gen = synth.PlaneStream(seed=1234); offs = [3, 57]
f0 = torch.from_numpy(np.stack([gen.frame(o + 1)[0] for o in offs])).cuda(); f1 = torch.from_numpy(np.stack([gen.frame(o + 2)[0] for o in offs])).cuda()
ex = ORBextractor(nfeatures=1000, width=640, height=480, max_batch=S, lib=lib); cap = ex.capacity
keys = torch.zeros((S, cap, 28), dtype=torch.uint8, device='cuda'); desc = torch.zeros((S, cap, 32), dtype=torch.uint8, device='cuda'); n = torch.zeros(S, dtype=torch.int32, device='cuda')
ex.extract_batch_dev(f1, 640, S, keys, desc, n); torch.cuda.synchronize(); nn = n.cpu().numpy()
fl = OpticalFlowLK(width=640, height=480, max_batch=S, lib=lib)
xy = torch.zeros((S, cap, 2), dtype=torch.float32, device='cuda'); status = torch.zeros((S, cap), dtype=torch.uint8, device='cuda')
k2 = torch.zeros_like(keys); d2 = torch.zeros_like(desc); n2 = torch.zeros_like(n)
def pyramids():
    return [fl.debug_level(slot, s, lv) for slot in (0, 1) for s in range(S) for lv in range(4)]
def run(co):
    torch.cuda.synchronize()
    if co: assert ext.corun_launch(blocks, iters, kind, 8, launches) == 0
    if victim == 'lk':
        fl.reset(); fl.lk_batch_dev(f0, 640, S, None, None, cap, None, None, stream=sV); fl.lk_batch_dev(f1, 640, S, keys, n, cap, xy, status, stream=sV)
        torch.cuda.synchronize()
        return [np.concatenate([xy[s, :nn[s]].cpu().numpy().view(np.uint32), status[s, :nn[s], None].cpu().numpy().astype(np.uint32)], 1) for s in range(S)], pyramids()
    k2.zero_(); d2.zero_(); torch.cuda.synchronize()
    if co: pass
    ex.extract_batch_dev(f1, 640, S, k2, d2, n2, stream=sV); torch.cuda.synchronize()
    return [np.concatenate([k2[s].cpu().numpy().reshape(cap, 28), d2[s].cpu().numpy().reshape(cap, 32)], 1).astype(np.uint32) for s in range(S)] + [n2.cpu().numpy().astype(np.uint32).reshape(1, -1)], []
ref, pref = run(False); again, _ = run(False)
print('quiet vs quiet identical:', all((a == b).all() for a, b in zip(ref, again)))
bad_runs = 0; bad_items = 0; bad_pyr = 0
for r in range(reps):
    o, p = run(True)
    d = sum(int((a != b).any(1).sum()) for a, b in zip(o, ref)); bad_items += d; bad_runs += d > 0
    bad_pyr += sum(int((a != b).sum()) for a, b in zip(p, pref))
print('victim %s mask %s co-runner kind %d blocks %d iters %d x %d: %d of %d runs differ, %d items (keypoints / rows) in total, pyramid bytes differing %d' % (victim, mask, kind, blocks, iters, launches, bad_runs, reps, bad_items, bad_pyr))
