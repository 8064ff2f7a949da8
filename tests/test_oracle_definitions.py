"""The oracle's restated third-party primitives against INDEPENDENT statements of their mathematical definitions (numpy / scipy / torch float64, nothing shared with oracle/*.c).
This does not pin OpenCV's bits (absent here: DESIGN.md §2 — parity stays unpinned); it pins what each primitive MEANS, so that a mis-restated alignment, border rule, tap set,
arc length or score definition cannot hide behind oracle == device agreement.  Fixed-point primitives must be within their own rounding of the exact real-valued result; the
integer-exact ones must be equal."""
import numpy as np
import pytest
from scipy import ndimage


def _img(seed, h=97, w=131):
    rng = np.random.RandomState(seed)
    base = ndimage.gaussian_filter(rng.rand(h, w), 2.0) * 900 - 200 + rng.rand(h, w) * 60
    return np.clip(base, 0, 255).astype(np.uint8)


@pytest.mark.parametrize('seed,dw,dh', [(1, 109, 81), (2, 110, 80), (3, 66, 49)])
def test_resize_linear_is_pixel_centre_bilinear(oracle, seed, dw, dh):
    """cv::resize(INTER_LINEAR): sample position (dx + 0.5) * sw / dw - 0.5, clamped at the borders, bilinear weights (11-bit fixed point in the restatement): against torch's float64
    bilinear interpolation with align_corners=False (the same definition) the fixed-point result is within 1 grey level everywhere and equal almost everywhere."""
    import torch
    import torch.nn.functional as F
    src = _img(seed)
    got = oracle.resize_linear(src, dw, dh).astype(np.int32)
    ref = F.interpolate(torch.from_numpy(src.astype(np.float64))[None, None], size=(dh, dw), mode='bilinear', align_corners=False)[0, 0].numpy()
    d = np.abs(got - ref)
    assert d.max() <= 1.0 + 1e-9 and (np.abs(got - np.rint(ref)) == 0).mean() > 0.9, (d.max(), (np.abs(got - np.rint(ref)) == 0).mean())


def test_gaussian7_is_sigma2_gaussian_reflect101(oracle):
    """cv::GaussianBlur(7 x 7, sigma 2, BORDER_REFLECT_101): against the exact normalised Gaussian taps exp(-x^2 / 8) convolved in float64 with scipy's 'mirror' border (= REFLECT_101)
    the 8.8 fixed-point result ({18, 34, 48, 56, 48, 34, 18} / 256 for {17.96, 33.55, 48.82, 55.33, ...}) stays within 1.5 grey levels, mean error below 0.3, borders included."""
    src = _img(4)
    x = np.arange(-3, 4); k = np.exp(-x * x / 8.0); k /= k.sum()
    ref = ndimage.convolve1d(ndimage.convolve1d(src.astype(np.float64), k, axis=0, mode='mirror'), k, axis=1, mode='mirror')
    got = oracle.gaussian7(src).astype(np.float64)
    d = np.abs(got - ref)
    assert d.max() <= 1.5 and d.mean() < 0.3, (d.max(), d.mean())
    assert np.abs((got - ref)[:3]).max() <= 1.5 and np.abs((got - ref)[:, -3:]).max() <= 1.5


def test_pyr_down_is_binomial5_decimation(oracle):
    """cv::pyrDown: [1 4 6 4 1]^2 / 256, REFLECT_101, every second sample, round half up — exact integers, so the independent statement must agree bit for bit"""
    for seed, (h, w) in ((5, (96, 128)), (6, (97, 131))):
        src = _img(seed, h, w)
        k = np.array([1, 4, 6, 4, 1], np.int64)
        full = ndimage.convolve1d(ndimage.convolve1d(src.astype(np.int64), k, axis=0, mode='mirror'), k, axis=1, mode='mirror')
        ref = ((full[::2, ::2] + 128) >> 8).astype(np.uint8)
        got = oracle.pyr_down(src)
        assert got.shape == ref.shape and (got == ref).all()


def test_scharr_deriv_is_3_10_3_central_difference(oracle):
    """calcSharrDeriv: Ix = [3 10 3]^T x [-1 0 1], Iy = [-1 0 1]^T x [3 10 3] as int16 (interior; the border rule is checked by the LK tests)"""
    src = _img(7).astype(np.int64)
    sm = np.array([3, 10, 3], np.int64); df = np.array([-1, 0, 1], np.int64)
    ix = ndimage.correlate1d(ndimage.correlate1d(src, sm, axis=0, mode='mirror'), df, axis=1, mode='mirror')
    iy = ndimage.correlate1d(ndimage.correlate1d(src, df, axis=0, mode='mirror'), sm, axis=1, mode='mirror')
    d = oracle.scharr_deriv(src.astype(np.uint8)).astype(np.int64)
    assert (d[1:-1, 1:-1, 0] == ix[1:-1, 1:-1]).all() and (d[1:-1, 1:-1, 1] == iy[1:-1, 1:-1]).all()


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _fast_score_definition(img):
    """S(p) = max over the 16 arcs of 9 contiguous ring pixels and both polarities of min over the arc of (v - ring) [darker ring] or (ring - v) [brighter ring], minus 1, floored at
    -1: p is a FAST-9/16 corner at threshold t iff some arc is entirely brighter than v + t or entirely darker than v - t iff S(p) >= t (cv::FAST's segment test and cornerScore)."""
    h, w = img.shape; v = img.astype(np.int64)
    ring = np.stack([np.roll(np.roll(v, -dy, 0), -dx, 1) for dx, dy in RING])          # ring[k][y, x] = img[y + dy, x + dx]
    dark = v[None] - ring; bright = ring - v[None]
    best = np.full((h, w), -10 ** 9, np.int64)
    for s in range(16):
        idx = [(s + j) % 16 for j in range(9)]
        best = np.maximum(best, np.maximum(dark[idx].min(0), bright[idx].min(0)))
    return best - 1


@pytest.mark.parametrize('seed', [11, 12])
def test_fast_is_the_9_of_16_segment_test_with_the_arc_score(oracle, seed):
    """cv::FAST(threshold, nonmax = false) returns exactly the interior pixels whose definition score is >= threshold, with that score as response, for both thresholds the
    extractor uses; with nonmax = true exactly those of them that are strictly greater than their 8 neighbours' scores (non-corners count 0)"""
    rng = np.random.RandomState(seed)
    img = _img(seed, 64, 80)
    img[rng.randint(4, 60, 60), rng.randint(4, 76, 60)] = rng.randint(0, 256, 60)          # isolated spikes: plenty of corners
    S = _fast_score_definition(img)
    inner = np.zeros_like(img, bool); inner[3:-3, 3:-3] = True
    for t in (20, 7):
        x, y, s = oracle.fast(img, t, nonmax=False)
        ref = inner & (S >= t)
        got = np.zeros_like(ref); got[y, x] = True
        assert (got == ref).all(), t
        assert (s == S[y, x]).all(), t
        xs, ys, ss = oracle.fast(img, t, nonmax=True)
        Sc = np.where(ref, S, 0)
        pad = np.pad(Sc, 1)
        nb = np.max(np.stack([pad[1 + dy:1 + dy + img.shape[0], 1 + dx:1 + dx + img.shape[1]] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)]), 0)
        refn = ref & (Sc > nb)
        gotn = np.zeros_like(ref); gotn[ys, xs] = True
        assert (gotn == refn).all() and refn.sum() > 10, t


def test_fast_atan2_against_arctan2(oracle):
    rng = np.random.RandomState(3)
    for _ in range(400):
        y, x = (float(v) for v in rng.uniform(-1000, 1000, 2))
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        d = abs(oracle.fast_atan2(y, x) - ref)
        assert min(d, 360 - d) < 0.02


def test_ic_angle_is_the_intensity_centroid_direction(oracle):
    """IC_Angle: atan2(m01, m10) of the radius-15 disc of half-widths umax[|v|] around the keypoint, in degrees"""
    umax = oracle.orb_params()['umax']
    img = _img(21, 80, 90)
    for (x, y) in ((40, 40), (33, 47), (60, 25)):
        m10 = m01 = 0
        for v in range(-15, 16):
            for u in range(-umax[abs(v)], umax[abs(v)] + 1):
                p = int(img[y + v, x + u]); m10 += u * p; m01 += v * p
        ref = np.degrees(np.arctan2(m01, m10)) % 360.0
        d = abs(oracle.ic_angle(img, x, y, umax) - ref)
        assert min(d, 360 - d) < 0.02


def _bilinear(img, x, y):
    """float64 bilinear sample with replicated borders"""
    h, w = img.shape
    x0 = np.floor(x).astype(int); y0 = np.floor(y).astype(int); fx = x - x0; fy = y - y0
    c = lambda a, lo, hi: np.clip(a, lo, hi)
    p = lambda yy, xx: img[c(yy, 0, h - 1), c(xx, 0, w - 1)]
    return (1 - fy) * ((1 - fx) * p(y0, x0) + fx * p(y0, x0 + 1)) + fy * ((1 - fx) * p(y0 + 1, x0) + fx * p(y0 + 1, x0 + 1))


def _lk_textbook(I, J, pts, levels=4, win=21, iters=30, eps=0.01):
    """Pyramidal Lucas-Kanade in float64, straight from the definition (Bouguet): binomial-5 pyramids, Scharr gradients / 32 of the first image, the 21 x 21 window sampled
    bilinearly, Gauss-Newton steps d = G^-1 b with G = sum [Ix^2 IxIy; IxIy Iy^2], b = sum (I - J) [Ix; Iy], from the coarsest level down.  No fixed point, no integer tricks."""
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16
    def down(a):
        f = ndimage.convolve1d(ndimage.convolve1d(a, k, axis=0, mode='mirror'), k, axis=1, mode='mirror')
        return f[::2, ::2]
    PI = [I.astype(np.float64)]; PJ = [J.astype(np.float64)]
    for _ in range(levels - 1): PI.append(np.floor(down(PI[-1]) + 0.5)); PJ.append(np.floor(down(PJ[-1]) + 0.5))       # the u8 pyramid of the library
    sm = np.array([3, 10, 3], np.float64) / 32; df = np.array([-1, 0, 1], np.float64)
    out = []
    half = (win - 1) / 2
    wy, wx = np.mgrid[0:win, 0:win].astype(np.float64)
    for (px, py) in pts:
        g = np.zeros(2)
        for l in range(levels - 1, -1, -1):
            A, B = PI[l], PJ[l]
            Ix = ndimage.correlate1d(ndimage.correlate1d(A, sm, axis=0, mode='mirror'), df, axis=1, mode='mirror')
            Iy = ndimage.correlate1d(ndimage.correlate1d(A, df, axis=0, mode='mirror'), sm, axis=1, mode='mirror')
            cx, cy = px / 2 ** l, py / 2 ** l
            X = cx - half + wx; Y = cy - half + wy
            a = _bilinear(A, X, Y); ix = _bilinear(Ix, X, Y); iy = _bilinear(Iy, X, Y)
            G = np.array([[np.sum(ix * ix), np.sum(ix * iy)], [np.sum(ix * iy), np.sum(iy * iy)]])
            v = np.zeros(2)
            for _ in range(iters):
                b_img = _bilinear(B, X + g[0] + v[0], Y + g[1] + v[1])
                e = b_img - a
                bb = np.array([np.sum(e * ix), np.sum(e * iy)])
                d = -np.linalg.solve(G, bb)
                v += d
                if d @ d <= eps * eps: break
            g = (g + v) * (2 if l > 0 else 1)
        out.append((px + g[0], py + g[1]))
    return np.array(out)


def test_lk_agrees_with_a_textbook_float64_lucas_kanade(oracle):
    """calcOpticalFlowPyrLK as restated (integer pyramids, 5-bit derivative scale, 14-bit bilinear weights, exact-sum accumulation) against a float64 Gauss-Newton LK written from
    the definition: on textured points of a warped frame pair both land within 0.01 px of each other (measured 4e-5 px median) and within 0.2 px of the true displacement — the fixed-point pipeline
    computes Lucas-Kanade flow, not something else that merely agrees with the device"""
    from sg_slam_amd import synth
    S = synth.PlaneStream(seed=1234)
    cur, prev = S.frame(11)[0], S.frame(10)[0]
    k, _ = oracle.orb_extract(cur)
    lvl0 = k[k['octave'] == 0]
    pts = np.stack([lvl0['x'], lvl0['y']], 1).astype('f4')
    pts = pts[(pts[:, 0] > 60) & (pts[:, 0] < 580) & (pts[:, 1] > 60) & (pts[:, 1] < 420)][:40]
    got, st = oracle.lk_pyr(cur, prev, pts, acc_mode=1)
    ref = _lk_textbook(cur, prev, pts.astype(np.float64))
    ok = st > 0
    d = np.linalg.norm(got[ok].astype(np.float64) - ref[ok], axis=1)
    assert ok.sum() >= 30 and np.median(d) < 2e-3 and d.max() < 1e-2, (ok.sum(), np.median(d), d.max())          # measured: median 4e-5 px, max 2e-4 px over 40 points
    A = synth.flow_affine(S, 11, 10)                                  # the true map cur -> prev on the plane
    truth = (A @ np.c_[pts.astype(np.float64), np.ones(len(pts))].T).T
    assert np.median(np.linalg.norm(got[ok].astype(np.float64) - truth[ok], axis=1)) < 0.2


@pytest.mark.parametrize('n,seed', [(400, 43), (800, 47)])
def test_pose_optimization_returns_the_least_squares_pose_of_its_inliers(oracle, n, seed):
    """Optimizer::PoseOptimization's last rounds run without the robust kernel on the edges still classified as inliers, so the pose it returns must be the weighted least-squares
    optimum of the reprojection error over exactly those edges — checked with scipy's own Levenberg-Marquardt (MINPACK) on an independently written residual (mono: 2 rows per
    edge, stereo: 3 with u_R = u - bf / z; weight 1 / sigma^2 of the octave): started from the returned pose it must not move (< 1e-5) nor lower chi^2 (< 1e-6 relative)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation as Rot
    from scenes import make_pose_problem, CAM
    frame, Ttrue, _ = make_pose_problem(oracle, n=n, seed=seed)
    is2 = oracle.orb_params()['inv_sigma2']
    ninl, T, out = oracle.pose_optimization(frame, CAM, is2)
    T = T.astype(np.float64)
    sel = (frame['has_mp'] > 0) & (out == 0)
    k = frame['keys']; X = frame['xw'].astype(np.float64)[sel]
    u = k['x'].astype(np.float64)[sel]; v = k['y'].astype(np.float64)[sel]; ur = frame['uright'].astype(np.float64)[sel]
    w = np.sqrt(is2[k['octave']].astype(np.float64)[sel]); stereo = ur >= 0

    def resid(p):
        R = Rot.from_rotvec(p[:3]).as_matrix() @ T[:3, :3]; t = T[:3, 3] + p[3:]
        Xc = X @ R.T + t
        pu = CAM['fx'] * Xc[:, 0] / Xc[:, 2] + CAM['cx']; pv = CAM['fy'] * Xc[:, 1] / Xc[:, 2] + CAM['cy']; pr = pu - CAM['bf'] / Xc[:, 2]
        return np.concatenate([w * (u - pu), w * (v - pv), (w * (ur - pr))[stereo]])

    c0 = float((resid(np.zeros(6)) ** 2).sum())
    sol = least_squares(resid, np.zeros(6), method='lm', xtol=1e-14, ftol=1e-14)
    assert ninl == sel.sum() and ninl > 0.5 * n
    assert np.abs(sol.x).max() < 1e-5 and c0 - float((sol.fun ** 2).sum()) < 1e-6 * c0, (np.abs(sol.x).max(), c0, float((sol.fun ** 2).sum()))
    assert np.abs(T - Ttrue).max() < 5e-3


def test_local_bundle_adjustment_ends_at_the_least_squares_optimum_of_its_kept_edges(oracle):
    """Optimizer::LocalBundleAdjustment's second pass (10 LM iterations, no robust kernel, erased edges excluded) on a LocalBA-sized graph: scipy's trust-region least squares on an
    independently written residual over the free poses and the points of the kept edges, started from the returned estimate, gains < 1e-5 of chi^2 and moves no pose by 1e-4
    (weakly observable point depths may still slide along their rays: not asserted)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation as Rot
    from scenes import make_ba_problem, CAM
    prob, _, _ = make_ba_problem(oracle, n_free=4, n_fixed=3, n_points=120, seed=11)
    poses, pts, erase, trace, iters = oracle.local_ba(prob, CAM)
    poses = poses.astype(np.float64); pts = pts.astype(np.float64)
    free = np.nonzero(prob['pose_fixed'] == 0)[0]
    keep = erase == 0
    ep, el = prob['edge_pose'][keep], prob['edge_point'][keep]
    eo, ei = prob['edge_obs'].astype(np.float64)[keep], prob['edge_info'].astype(np.float64)[keep]
    used = np.unique(el)
    w = np.sqrt(ei); stereo = eo[:, 2] >= 0

    def resid(p):
        dp = p[:6 * len(free)].reshape(-1, 6); dx = p[6 * len(free):].reshape(-1, 3)
        R = poses[:, :3, :3].copy(); t = poses[:, :3, 3].copy()
        for j, i in enumerate(free):
            R[i] = Rot.from_rotvec(dp[j, :3]).as_matrix() @ poses[i, :3, :3]; t[i] = poses[i, :3, 3] + dp[j, 3:]
        X = pts.copy(); X[used] += dx
        Xc = np.einsum('eij,ej->ei', R[ep], X[el]) + t[ep]
        pu = CAM['fx'] * Xc[:, 0] / Xc[:, 2] + CAM['cx']; pv = CAM['fy'] * Xc[:, 1] / Xc[:, 2] + CAM['cy']; pr = pu - CAM['bf'] / Xc[:, 2]
        return np.concatenate([w * (eo[:, 0] - pu), w * (eo[:, 1] - pv), (w * (eo[:, 2] - pr))[stereo]])

    n0 = 6 * len(free) + 3 * len(used)
    c0 = float((resid(np.zeros(n0)) ** 2).sum())
    assert abs(c0 - trace[1, iters[1] - 1, 0]) <= 1e-6 * c0                 # the independent residual reproduces the oracle's own final chi^2 (float poses / points rounded at the boundary)
    sol = least_squares(resid, np.zeros(n0), method='trf', xtol=1e-15, ftol=1e-15, gtol=1e-15)
    c1 = float((sol.fun ** 2).sum())
    assert (c0 - c1) < 1e-5 * c0 and np.abs(sol.x[:6 * len(free)]).max() < 1e-4, (c0, c1, np.abs(sol.x[:6 * len(free)]).max())


def _rot_cw(img):
    """rotate the image content by +90 degrees in image coordinates (x to the right, y down; the sense in which IC_Angle / KeyPoint::angle grow): N[y', x'] = B[y, x] with
    (x', y') = (H - 1 - y, x)"""
    return np.ascontiguousarray(np.rot90(img, k=-1))


def test_orientation_and_descriptor_are_rotation_equivariant(oracle):
    """Geometric meaning of the steering: rotating the image by +90 degrees turns the intensity-centroid angle by +90 degrees and leaves the rBRIEF descriptor computed at the
    rotated keypoint with the rotated angle UNCHANGED (angles 0 / 90 / 180 / 270: cos / sin are 0, +-1 up to 4e-8, so the rounded sample offsets are the exactly rotated ones);
    FAST corners and scores rotate with the image."""
    umax = oracle.orb_params()['umax']
    img = _img(31, 101, 101)                                       # square: H - 1 - y stays inside
    blur = oracle.gaussian7(img)
    assert (oracle.gaussian7(_rot_cw(img)) == _rot_cw(blur)).all()                 # the symmetric blur commutes with the rotation
    H = img.shape[0]
    for (x, y) in ((50, 50), (40, 57), (61, 44)):
        imgs = [img]; blurs = [blur]; pos = [(x, y)]
        for _ in range(3):
            imgs.append(_rot_cw(imgs[-1])); blurs.append(_rot_cw(blurs[-1])); px, py = pos[-1]; pos.append((H - 1 - py, px))
        a0 = oracle.ic_angle(img, x, y, umax)
        d0 = oracle.descriptor(blur, x, y, 0.0)
        for q in range(1, 4):
            aq = oracle.ic_angle(imgs[q], pos[q][0], pos[q][1], umax)
            dd = (aq - a0 - 90.0 * q) % 360.0
            assert min(dd, 360.0 - dd) < 0.02, (q, a0, aq)
            assert (oracle.descriptor(blurs[q], pos[q][0], pos[q][1], 90.0 * q) == d0).all(), q
        assert d0.any()
    x0, y0, s0 = oracle.fast(img, 20, nonmax=True)
    x1, y1, s1 = oracle.fast(_rot_cw(img), 20, nonmax=True)
    a = sorted(zip((H - 1 - y0).tolist(), x0.tolist(), s0.tolist())); b = sorted(zip(x1.tolist(), y1.tolist(), s1.tolist()))
    assert a == b and len(a) > 5


def test_rgbd_stereo_and_unprojection_are_the_pinhole_model(oracle):
    """ComputeStereoFromRGBD: z = raw / 5000 at the keypoint's TRUNCATED pixel (imDepth.at<float>(v, u) with float arguments, Frame.cc:903-906), u_R = u - bf / z; UnprojectStereo: the world point that the pose + pinhole model project back onto the
    keypoint with that depth (float32 round trip)"""
    from sg_slam_amd import synth
    from scenes import CAM
    S = synth.LayeredStream(seed=1234)
    g, depth, T = S.frame(7)
    k, _ = oracle.orb_extract(g)
    ur, z = oracle.compute_stereo_from_rgbd(k, depth, CAM['bf'], CAM['depth_factor'])
    xi = np.floor(k['x'].astype(np.float64)).astype(int); yi = np.floor(k['y'].astype(np.float64)).astype(int)
    zr = depth[np.clip(yi, 0, 479), np.clip(xi, 0, 639)].astype(np.float64) / CAM['depth_factor']
    good = z > 0
    assert good.mean() > 0.9 and np.abs(z[good] - zr[good]).max() < 1e-6 and np.abs(ur[good] - (k['x'][good] - CAM['bf'] / zr[good])).max() < 1e-3
    xw, has = oracle.unproject_stereo(k, z, T.astype('f4'), CAM)
    assert (has.astype(bool) == good).all()
    Xc = xw[good].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    u = CAM['fx'] * Xc[:, 0] / Xc[:, 2] + CAM['cx']; v = CAM['fy'] * Xc[:, 1] / Xc[:, 2] + CAM['cy']
    assert np.abs(u - k['x'][good]).max() < 2e-3 and np.abs(v - k['y'][good]).max() < 2e-3 and np.abs(Xc[:, 2] - z[good]).max() < 1e-5


def test_detector_preprocessing_is_pixel_centre_bilinear_minus_mean():
    """ncnn::Mat::from_pixels_resize (fixed-point bilinear, as restated in oracle/detector_oracle.py) + substract_mean_normalize: within one grey level of torch's float64
    bilinear interpolation with align_corners = False, channel means removed"""
    import torch
    import torch.nn.functional as F
    from oracle import detector_oracle as D
    rng = np.random.RandomState(5)
    img = np.clip(ndimage.gaussian_filter(rng.rand(480, 640, 3), (2, 2, 0)) * 700 - 150, 0, 255).astype(np.uint8)
    x = D.preprocess(img)
    ref = F.interpolate(torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None], size=(300, 300), mode='bilinear', align_corners=False)[0].numpy() - D.MEAN.astype(np.float64)[:, None, None]
    assert x.shape == (3, 300, 300) and np.abs(x - ref).max() <= 1.0 + 1e-4


def test_sim3_exp_is_the_matrix_exponential_and_log_inverts_it(oracle):
    """g2o::Sim3(update) (types/sim3.h: omega, upsilon, sigma -> r = exp(omega), s = e^sigma, t = W upsilon) against scipy's matrix exponential of the 4 x 4 generator
    [[omega^ + sigma I, upsilon], [0, 0]] = [[s R, t], [0, 1]], and Sim3::log as its inverse — the group the loop-closing optimisers (OptimizeSim3, OptimizeEssentialGraph) move in"""
    import ctypes as C
    from scipy.linalg import expm
    from scipy.spatial.transform import Rotation as Rot
    L = oracle.lib()
    rng = np.random.RandomState(9)
    for scale in (1e-3, 0.1, 1.0, 2.5):
        for _ in range(6):
            u = rng.randn(7) * scale; u[6] = rng.randn() * min(scale, 0.5)
            out = np.zeros(8)
            L.orc_kat_sim3_exp(u.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
            w = u[:3]
            G = np.zeros((4, 4)); G[:3, :3] = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) + u[6] * np.eye(3); G[:3, 3] = u[3:6]
            M = expm(G)
            s = out[7]; R = Rot.from_quat(out[:4]).as_matrix()
            assert abs(s - np.exp(u[6])) < 1e-12 * max(1, s)
            assert np.abs(s * R - M[:3, :3]).max() < 1e-9 * max(1.0, np.abs(M).max()) and np.abs(out[4:7] - M[:3, 3]).max() < 1e-9 * max(1.0, np.abs(M).max()), (scale, u)
            back = np.zeros(7)
            L.orc_kat_sim3_log(out.ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p))
            th = np.linalg.norm(w)
            if 0.01 < th < 3.0:                                   # the general branch of Sim3::log (cos(theta) <= 1 - 1e-5) inside the principal branch of the rotation logarithm
                assert np.abs(back - u).max() < 1e-8 * max(1.0, np.abs(u).max()), (scale, u, back)
            elif th < 4e-3 and abs(u[6]) >= 1e-5:
                # REFERENCE QUIRK, reproduced on purpose: for cos(theta) > 1 - 1e-5 (theta < 4.5e-3) and |sigma| >= 1e-5, sim3.h:195-199 sets
                # B = ((sigma^2 / 2 - sigma + 1) s) / sigma^3 — the "- 1" of the series is missing, B ~ 1 / sigma^3 instead of ~ 1 / 6 — so log() does NOT invert exp() there:
                # the translation part comes back wrong.  (Sim3(update) has the same formula, but only below theta = 1e-5.)  The oracle must follow the reference, not the mathematics.
                sg = u[6]; sc = np.exp(sg)
                A_ = ((sg - 1) * sc + 1) / sg ** 2; B_ = ((0.5 * sg * sg - sg + 1) * sc) / sg ** 3; C_ = (sc - 1) / sg
                Rm = Rot.from_quat(out[:4]).as_matrix(); om = 0.5 * np.array([Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]])
                Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
                ups = np.linalg.solve(A_ * Om + B_ * Om @ Om + C_ * np.eye(3), out[4:7])
                assert np.abs(back[3:6] - ups).max() < 1e-6 * max(1e-3, np.abs(ups).max()) and np.abs(back[:3] - om).max() < 1e-12 and abs(back[6] - sg) < 1e-12
                assert np.abs(back[3:6] - u[3:6]).max() > 1e-3 * np.abs(u[3:6]).max()                 # ... and that is not the inverse of exp


def test_detection_output_against_an_independent_vectorised_nms():
    """ncnn DetectionOutput as restated in oracle/detector_oracle.py against a second implementation written differently (decoded boxes and the full IoU matrix by numpy
    broadcasting in float32, greedy suppression as a boolean sweep): same rows on random head outputs with heavy overlaps"""
    from oracle import detector_oracle as D
    from test_detector import PARAM
    layers = D.parse_param(PARAM)
    p = [L for L in layers if L['type'] == 'DetectionOutput'][0]['p']
    pri = [D.prior_box(fh, fh, 300, L['p']) for L, fh in zip([L for L in layers if L['type'] == 'PriorBox'], (19, 10, 5, 3, 2, 1))]
    priors = np.concatenate(pri, 1)
    n = priors.shape[1] // 4; nc = p[0]
    rng = np.random.RandomState(4)
    for trial in range(3):
        loc = (rng.randn(n, 4) * (0.3, 1.0, 2.0)[trial]).astype(np.float32)
        raw = rng.randn(n, nc).astype(np.float32) * 2; raw[:, 0] += 2
        conf = np.exp(raw - raw.max(1, keepdims=True)); conf = (conf / conf.sum(1, keepdims=True)).astype(np.float32)
        ref = D.detection_output(loc.reshape(-1), conf.reshape(-1), priors, p)
        # independent implementation
        var = np.array([p.get(5, .1), p.get(6, .1), p.get(7, .2), p.get(8, .2)], np.float32)
        pb = priors[0].reshape(-1, 4)
        pw = pb[:, 2] - pb[:, 0]; ph = pb[:, 3] - pb[:, 1]; pcx = (pb[:, 0] + pb[:, 2]) * np.float32(.5); pcy = (pb[:, 1] + pb[:, 3]) * np.float32(.5)
        cx = var[0] * loc[:, 0] * pw + pcx; cy = var[1] * loc[:, 1] * ph + pcy
        w = np.exp(var[2] * loc[:, 2]).astype(np.float32) * pw; h = np.exp(var[3] * loc[:, 3]).astype(np.float32) * ph
        B = np.stack([cx - w * np.float32(.5), cy - h * np.float32(.5), cx + w * np.float32(.5), cy + h * np.float32(.5)], 1).astype(np.float32)
        rows = []
        for c in range(1, nc):
            s = conf[:, c]; cand = np.nonzero(s > np.float32(p[4]))[0]
            cand = cand[np.lexsort((cand, -s[cand]))][:p[2]]                      # score descending, prior index ascending
            b = B[cand]; area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
            iw = np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0])
            ih = np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1])
            inter = np.where((iw > 0) & (ih > 0), (iw * ih).astype(np.float32), np.float32(0))
            iou = inter / (area[:, None] + area[None, :] - inter)
            alive = np.ones(len(cand), bool); kept = []
            for i in range(len(cand)):
                if not alive[i]: continue
                kept.append(i); alive &= ~(iou[i] > np.float32(p[1])); alive[i] = False
            rows += [(c, s[cand[i]], *b[i]) for i in kept]
        rows.sort(key=lambda r: -r[1])
        mine = np.array(rows[:p[3]], np.float32).reshape(-1, 6)
        assert mine.shape == ref.shape and len(ref) == p[3] and (mine == ref).all(), trial
