import sys, os, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, ctypes as C
import sg_slam_amd
from sg_slam_amd.capi import _vp, KP_DTYPE
from sg_slam_amd.matcher import camera_struct
from oracle import oracle as orc
from scenes import make_pose_problem, CAM
from _campaign_lib import taps_lib
lib = taps_lib()          # the tap build (include/sgx_debug.h): plan selection / per-step timing / blob read-back are not in the product library
is2 = np.asarray(orc.orb_params()['inv_sigma2'], 'f4')
fr, _, _ = make_pose_problem(orc, n=800, seed=44)
cap = 1024; n = len(fr['keys'])
cs = camera_struct(CAM)
for S in (64, 256, 512, 1024):
    keys = np.zeros((S, cap), KP_DTYPE); keys[:, :n] = fr['keys']
    ur = np.zeros((S, cap), 'f4'); ur[:, :n] = fr['uright']
    has = np.zeros((S, cap), np.uint8); has[:, :n] = fr['has_mp']
    xw = np.zeros((S, cap, 3), 'f4'); xw[:, :n] = fr['xw']
    T0 = np.tile(np.asarray(fr['Tcw'], 'f4').reshape(1, 16), (S, 1))
    d = lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).cuda()
    dk, du, dh, dx = d(keys), d(ur), d(has), d(xw); dn = torch.full((S,), n, dtype=torch.int32, device='cuda')
    out = torch.zeros((S, cap), dtype=torch.uint8, device='cuda'); ninl = torch.zeros((S,), dtype=torch.int32, device='cuda')
    for thr in (256, 64):
        lib.tap('sgx_pose_opt_debug_set_threads')(thr)
        def run():
            dT = torch.from_numpy(T0).cuda()
            lib.check(lib.dll.sgx_pose_optimization_batch_dev(S, cap, _vp(dk), _vp(du), _vp(dn), None, _vp(dh), _vp(dx), cap, _vp(is2), len(is2), C.byref(cs), _vp(dT), _vp(out), _vp(ninl), None))
        for _ in range(3): run()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print(json.dumps(dict(S=S, threads=thr, ms=dt * 1e3, frames_per_s=S / dt)))
