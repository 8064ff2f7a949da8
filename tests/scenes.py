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


def make_ba_problem(orc, n_free=8, n_fixed=5, n_points=400, seed=11, outlier_frac=0.08, noise_px=0.8, mono_frac=0.2, pose_sigma=0.01, point_sigma=0.03):
    """SURVEY.md §8(d) input 4, LocalBA-sized: keyframes on a short arc looking at a point cloud, every point seen by 3..8
    keyframes, pixel noise sigma*level scale, gross outliers, initial poses/points perturbed.  poses[0] is the
    mnId==0 keyframe (pose_fixed 2), then n_free local keyframes (0), then n_fixed fixed cameras (1)."""
    rng = np.random.RandomState(seed)
    p = orc.orb_params()
    npose = 1 + n_free + n_fixed
    Ts = []
    for i in range(npose):
        a = 0.04 * i
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        C = np.array([0.15 * i, 0.02 * np.sin(i), 0.01 * i])
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = -R @ C
        Ts.append(T)
    Ts = np.stack(Ts)
    pts = np.stack([rng.uniform(-1.5, 3.0, n_points), rng.uniform(-1.2, 1.2, n_points), rng.uniform(2.0, 6.0, n_points)], 1)
    fixed = np.array([2] + [0] * n_free + [1] * n_fixed, np.uint8)
    ep, el, eo, ei = [], [], [], []
    for l in range(n_points):
        seen = rng.choice(npose, size=min(rng.randint(3, 9), npose), replace=False)
        for i in sorted(seen):                      # std::map<KeyFrame*,size_t> order stand-in
            Xc = Ts[i, :3, :3] @ pts[l] + Ts[i, :3, 3]
            if Xc[2] < 0.3:
                continue
            u = CAM['fx'] * Xc[0] / Xc[2] + CAM['cx']; v = CAM['fy'] * Xc[1] / Xc[2] + CAM['cy']
            if not (0 < u < 640 and 0 < v < 480):
                continue
            octv = rng.randint(0, 8); sig = noise_px * p['scale'][octv]
            uo = u + rng.randn() * sig; vo = v + rng.randn() * sig
            ur = uo - CAM['bf'] / Xc[2] + rng.randn() * sig * 0.5
            if rng.rand() < outlier_frac:
                uo += rng.uniform(-50, 50); vo += rng.uniform(-50, 50)
            if rng.rand() < mono_frac:
                ur = -1.0
            ep.append(i); el.append(l); eo.append([uo, vo, ur]); ei.append(p['inv_sigma2'][octv])
    # a map point enters the local graph with at least two observations (single-view points have unobservable depth)
    cnt = np.bincount(np.array(el, 'i8'), minlength=n_points)
    keep = [k for k in range(len(el)) if cnt[el[k]] >= 2]
    ep = [ep[k] for k in keep]; el = [el[k] for k in keep]; eo = [eo[k] for k in keep]; ei = [ei[k] for k in keep]
    poses0 = Ts.copy()
    for i in range(npose):
        if fixed[i] == 0:
            d = rng.randn(6) * pose_sigma
            dR = np.eye(3) + np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
            uu, _, vv = np.linalg.svd(dR); dR = uu @ vv
            poses0[i, :3, :3] = dR @ Ts[i, :3, :3]; poses0[i, :3, 3] = dR @ Ts[i, :3, 3] + d[3:]
    pts0 = pts + rng.randn(*pts.shape) * point_sigma
    return dict(poses=poses0.astype('f4'), pose_fixed=fixed, points=pts0.astype('f4'), edge_pose=np.array(ep, 'i4'), edge_point=np.array(el, 'i4'),
                edge_obs=np.array(eo, 'f4'), edge_info=np.array(ei, 'f4')), Ts, pts


def make_local_map(orc, S, t, seed=0, n_prev=2):
    """cur = frame t; local map = keypoints of frames t-1..t-n_prev unprojected with their true poses (MapPoint ctor,
    MapPoint.cc:44-67: normal = unit vector from the camera centre, max_dist = dist*scale[octave], min_dist = max_dist/scale[nlevels-1]).
    Some keypoints of cur already hold observed map points (cur['mp_obs'] > 0), some local points are skipped / unobserved."""
    rng = np.random.RandomState(seed)
    p = orc.orb_params(); sf = p['scale']
    g, d, T = S.frame(t)
    k, desc = orc.orb_extract(g)
    ur, z = orc.compute_stereo_from_rgbd(k, d, CAM['bf'], CAM['depth_factor'])
    cur = dict(keys=k, desc=desc, uright=ur, Tcw=T.astype('f4'))
    xs, ns, mind, maxd, ds = [], [], [], [], []
    for j in range(1, n_prev + 1):
        gj, dj, Tj = S.frame(t - j)
        kj, dsj = orc.orb_extract(gj)
        urj, zj = orc.compute_stereo_from_rgbd(kj, dj, CAM['bf'], CAM['depth_factor'])
        xw, has = orc.unproject_stereo(kj, zj, Tj.astype('f4'), CAM)
        Ow = -(Tj[:3, :3].T @ Tj[:3, 3])
        sel = has > 0
        PO = xw[sel] - Ow.astype('f4')
        dist = np.linalg.norm(PO.astype('f8'), axis=1).astype('f4')
        xs.append(xw[sel]); ns.append((PO / dist[:, None]).astype('f4'))
        mx = (dist * sf[kj['octave'][sel]]).astype('f4'); maxd.append(mx); mind.append((mx / sf[7]).astype('f4')); ds.append(dsj[sel])
    xw = np.concatenate(xs); n = len(xw)
    lm = dict(xw=xw, normal=np.concatenate(ns), min_dist=np.concatenate(mind), max_dist=np.concatenate(maxd), desc=np.concatenate(ds),
              obs=(rng.rand(n) < 0.7).astype('i4') * rng.randint(1, 5, n), skip=(rng.rand(n) < 0.1).astype(np.uint8))
    cur['mp_obs'] = np.where(rng.rand(len(k)) < 0.25, rng.randint(0, 3, len(k)), -1).astype('i4')
    return cur, lm


def make_big_ba_problem(nkf=2000, npt=50000, seed=4):
    """BASELINE config 4 generator (SURVEY.md §8(d) input 4): `nkf` keyframes on three loops of a 15 m circle looking outward, `npt` landmarks on walls
    3..8 m away, each seen from 5..11 consecutive keyframes; pixel noise per octave, 5 % gross outliers, 20 % mono observations, perturbed initial poses
    (the first one fixed) and points.  Returns (problem dict for Optimizer.LocalBundleAdjustment, true poses [nkf,4,4] f64, initial poses f64)."""
    NKF, NPT = nkf, npt
    rng = np.random.RandomState(seed)
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
    return prob, Ts, poses0
