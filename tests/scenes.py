"""Shared test scenes: consecutive synthetic RGB-D frames turned into the flattened Frame arrays the
matcher / pose optimiser consume (built with the oracle's extractor + stereo helpers)."""
import numpy as np
from sg_slam_amd import synth

CAM = dict(synth.TUM3)


def make_pair(orc, S, t, seed=0, obs_mode='mixed', pose_noise=0.0):
    """last = frame t, cur = frame t+1.  Every last-frame keypoint with depth carries a map point
    (UnprojectStereo with the true pose).  obs_mode: 'zero' (visual-odometry points, no locks),
    'mixed' (half observed -> exercises the greedy lock rule), 'all'."""
    rng = np.random.RandomState(seed)
    gl, dl, Tl = S.frame(t)
    gc, dc, Tc = S.frame(t + 1)
    kl, desl = orc.orb_extract(gl)
    kc, desc = orc.orb_extract(gc)
    url, zl = orc.compute_stereo_from_rgbd(kl, dl, CAM['bf'], CAM['depth_factor'])
    urc, zc = orc.compute_stereo_from_rgbd(kc, dc, CAM['bf'], CAM['depth_factor'])
    Tl32 = Tl.astype('f4')
    xw, has = orc.unproject_stereo(kl, zl, Tl32, CAM)
    n = len(kl)
    if obs_mode == 'zero':
        obs = np.zeros(n, 'i4')
    elif obs_mode == 'all':
        obs = np.full(n, 3, 'i4')
    else:
        obs = (rng.rand(n) < 0.5).astype('i4') * rng.randint(1, 6, n)
    outlier = (rng.rand(n) < 0.03).astype(np.uint8)
    mpdesc = desl.copy()
    flip = rng.rand(n) < 0.3                       # distinctive descriptors differ a little from the keypoint's own
    for i in np.nonzero(flip)[0]:
        b = rng.randint(0, 256, 6)
        mpdesc[i, b // 8] ^= (1 << (b % 8)).astype(np.uint8)
    Tc32 = Tc.astype('f4')
    if pose_noise:
        Tc32 = Tc32.copy(); Tc32[:3, 3] += rng.randn(3).astype('f4') * pose_noise
    last = dict(keys=kl, desc=desl, uright=url, zdepth=zl, Tcw=Tl32, has_mp=has, outlier=outlier, xw=xw, obs=obs, mpdesc=mpdesc)
    cur = dict(keys=kc, desc=desc, uright=urc, zdepth=zc, Tcw=Tc32)
    return cur, last


def make_pose_problem(orc, n=400, seed=42, outlier_frac=0.2, noise_px=1.0, mono_frac=0.15, init_sigma=0.02):
    """SURVEY.md §8(d) input 3: points in a 4 m frustum projected with TUM3 intrinsics, sigma = 1 px * level
    scale, gross outliers, stereo observations with bf = 40 (a fraction mono: mvuRight = -1), initial pose =
    truth o exp(N(0, init_sigma))."""
    from oracle.oracle import KP_DTYPE
    rng = np.random.RandomState(seed)
    p = orc.orb_params()
    th = rng.randn(3) * 0.2
    cz, sz = np.cos(th[2]), np.sin(th[2]); cy_, sy = np.cos(th[1]), np.sin(th[1]); cx_, sx = np.cos(th[0]), np.sin(th[0])
    R = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]]) @ np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]])
    t = rng.randn(3) * 0.5
    Ttrue = np.eye(4); Ttrue[:3, :3] = R; Ttrue[:3, 3] = t
    z = rng.uniform(0.5, 4.0, n)
    u = rng.uniform(20, 620, n); v = rng.uniform(20, 460, n)
    Xc = np.stack([(u - CAM['cx']) * z / CAM['fx'], (v - CAM['cy']) * z / CAM['fy'], z], 1)
    Xw = (R.T @ (Xc - t).T).T
    octave = rng.randint(0, 8, n)
    sig = noise_px * p['scale'][octave]
    un = u + rng.randn(n) * sig; vn = v + rng.randn(n) * sig
    ur = un - CAM['bf'] / z + rng.randn(n) * sig * 0.5
    gross = rng.rand(n) < outlier_frac
    un[gross] += rng.uniform(-60, 60, gross.sum()); vn[gross] += rng.uniform(-60, 60, gross.sum())
    mono = rng.rand(n) < mono_frac
    ur[mono] = -1.0
    keys = np.zeros(n, KP_DTYPE); keys['x'] = un; keys['y'] = vn; keys['octave'] = octave; keys['class_id'] = -1
    has = (rng.rand(n) < 0.9).astype(np.uint8)
    d = rng.randn(6) * init_sigma
    dR = np.eye(3) + np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
    uu, _, vv = np.linalg.svd(dR); dR = uu @ vv
    T0 = np.eye(4); T0[:3, :3] = dR @ R; T0[:3, 3] = dR @ t + d[3:]
    frame = dict(keys=keys, uright=ur.astype('f4'), has_mp=has, xw=Xw.astype('f4'), Tcw=T0.astype('f4'))
    return frame, Ttrue, gross
