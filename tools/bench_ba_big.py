"""BASELINE config 4: bundle adjustment at 2 000 keyframes / 50 000 landmarks (synthetic, SURVEY.md §8(d) input 4) through
sgx_local_bundle_adjustment (same edges / solver as Optimizer::BundleAdjustment, Optimizer.cc:49-237).  GPU only — the dense CPU oracle
is not run at this size; the check is the chi2 trajectory (must fall) and the recovered poses vs the generator's truth.
usage: python tools/bench_ba_big.py [n_kf] [n_points]   -> one JSON line"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sg_slam_amd
from sg_slam_amd.optimizer import Optimizer
from scenes import CAM

NKF = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
NPT = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
rng = np.random.RandomState(4)
# 3 loops of a 15 m circle, cameras looking outward (+z = radial direction), walls of landmarks 3..8 m away
th = 3 * 2 * np.pi * np.arange(NKF) / NKF
C = np.stack([15 * np.cos(th), 0.05 * np.sin(5 * th), 15 * np.sin(th)], 1) * (1 + 0.02 * np.arange(NKF)[:, None] / NKF)
zc = np.stack([np.cos(th), np.zeros(NKF), np.sin(th)], 1); yc = np.tile([0.0, 1.0, 0.0], (NKF, 1)); xc = np.cross(yc, zc)
R = np.stack([xc, yc, zc], 1)                                   # rows = camera axes in world coords: Xc = R (Xw - C)
Ts = np.tile(np.eye(4), (NKF, 1, 1)); Ts[:, :3, :3] = R; Ts[:, :3, 3] = -np.einsum('nij,nj->ni', R, C)
base = rng.randint(0, NKF, NPT); k = rng.randint(5, 12, NPT)
mid = (base + k // 2) % NKF
depth = rng.uniform(3.0, 8.0, NPT); lat = rng.uniform(-0.45, 0.45, NPT) * depth; up = rng.uniform(-0.3, 0.3, NPT) * depth
pts = C[mid] + zc[mid] * depth[:, None] + xc[mid] * lat[:, None] + yc[mid] * up[:, None]
sig = np.array([1.2 ** i for i in range(8)]); inv_sigma2 = 1.0 / sig ** 2
ep, el = [], []
for j in range(11):
    sel = np.nonzero(k > j)[0]; ep.append((base[sel] + j) % NKF); el.append(sel)
ep = np.concatenate(ep); el = np.concatenate(el)
o = np.lexsort((ep, el)); ep, el = ep[o], el[o]
Xc = np.einsum('nij,nj->ni', Ts[ep, :3, :3], pts[el]) + Ts[ep, :3, 3]
u = CAM['fx'] * Xc[:, 0] / Xc[:, 2] + CAM['cx']; v = CAM['fy'] * Xc[:, 1] / Xc[:, 2] + CAM['cy']
ok = (Xc[:, 2] > 0.3) & (u > 0) & (u < 640) & (v > 0) & (v < 480)
ep, el, Xc, u, v = ep[ok], el[ok], Xc[ok], u[ok], v[ok]
cnt = np.bincount(el, minlength=NPT); ok = cnt[el] >= 2
ep, el, Xc, u, v = ep[ok], el[ok], Xc[ok], u[ok], v[ok]
ne = len(ep)
octv = rng.randint(0, 8, ne); s = 0.8 * sig[octv]
uo = u + rng.randn(ne) * s; vo = v + rng.randn(ne) * s; ur = uo - CAM['bf'] / Xc[:, 2] + rng.randn(ne) * s * 0.5
out = rng.rand(ne) < 0.05; uo[out] += rng.uniform(-50, 50, out.sum()); vo[out] += rng.uniform(-50, 50, out.sum())
ur[rng.rand(ne) < 0.2] = -1.0
poses0 = Ts.copy()
d = rng.randn(NKF, 6) * 0.01; d[0] = 0
for i in range(1, NKF):
    w = d[i, :3]; dR = np.eye(3) + np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]); uu, _, vv = np.linalg.svd(dR); dR = uu @ vv
    poses0[i, :3, :3] = dR @ Ts[i, :3, :3]; poses0[i, :3, 3] = dR @ Ts[i, :3, 3] + d[i, 3:]
fixed = np.zeros(NKF, np.uint8); fixed[0] = 2
prob = dict(poses=poses0.astype('f4'), pose_fixed=fixed, points=(pts + rng.randn(NPT, 3) * 0.03).astype('f4'), edge_pose=ep.astype('i4'), edge_point=el.astype('i4'),
            edge_obs=np.stack([uo, vo, ur], 1).astype('f4'), edge_info=inv_sigma2[octv].astype('f4'))
if os.environ.get('SGX_TOOL_EMU'):          # generator / plumbing check without a GPU (tests' kernel-logic emulator; never a measurement)
    from sg_slam_amd.capi import SgxLib
    lib = SgxLib(os.path.join(ROOT, 'tests', 'emu', 'libsgx_emu.so'))
else:
    lib = sg_slam_amd.load()
t = time.perf_counter(); er, st = Optimizer.LocalBundleAdjustment(prob, CAM, lib=lib); dt = time.perf_counter() - t
err_before = np.abs(poses0[:, :3, 3] - Ts[:, :3, 3]).max(); err_after = np.abs(prob['poses'].astype('f8')[:, :3, 3] - Ts[:, :3, 3]).max()
its = int(sum(st['iterations']))
print(json.dumps(dict(bench='bundle_adjustment_big', keyframes=NKF, landmarks=NPT, edges=int(ne), reduced_system=6 * (NKF - 1), lm_iterations=its, seconds=dt,
                      edges_per_s=ne * its / dt, stats={k2: (list(map(float, v2)) if hasattr(v2, '__len__') else float(v2)) for k2, v2 in st.items()},
                      max_abs_translation_error_before=float(err_before), max_abs_translation_error_after=float(err_after))))
