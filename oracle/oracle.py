"""ctypes loader for oracle/liboracle.so (the CPU restatement of the reference path).
TEST INFRASTRUCTURE ONLY — never imported by the product package."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([('x', 'f4'), ('y', 'f4'), ('size', 'f4'), ('angle', 'f4'), ('response', 'f4'),
                     ('octave', 'i4'), ('class_id', 'i4')])
assert KP_DTYPE.itemsize == 28


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_fast_atan2.restype = C.c_float
        _LIB.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _LIB.orc_ic_angle.restype = C.c_float
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def orb_params(nfeatures=1000, scale=1.2, nlevels=8):
    L = lib()
    sc = np.zeros(nlevels, 'f4'); isc = np.zeros(nlevels, 'f4'); s2 = np.zeros(nlevels, 'f4'); is2 = np.zeros(nlevels, 'f4')
    per = np.zeros(nlevels, 'i4'); umax = np.zeros(16, 'i4')
    L.orc_orb_params_flat(C.c_int(nfeatures), C.c_float(scale), C.c_int(nlevels), _p(sc), _p(isc), _p(s2), _p(is2), _p(per), _p(umax))
    return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, per_level=per, umax=umax)


def level_sizes(w, h, scale=1.2, nlevels=8):
    p = orb_params(1000, scale, nlevels)
    out = []
    for l in range(nlevels):
        s = p['inv_scale'][l]
        out.append((int(np.rint(np.float32(w) * s)), int(np.rint(np.float32(h) * s))))
    return out


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), C.c_int(src.shape[1]), C.c_int(src.shape[0]), C.c_int(src.shape[1]),
                               _p(dst), C.c_int(dw), C.c_int(dh), C.c_int(dw))
    return dst


def gaussian7(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    h, w = src.shape
    lib().orc_gaussian7_u8(_p(src), C.c_int(w), C.c_int(h), C.c_int(w), _p(dst), C.c_int(w))
    return dst


def fast(img, threshold, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h
    ox = np.zeros(cap, 'i4'); oy = np.zeros(cap, 'i4'); os_ = np.zeros(cap, 'i4')
    n = lib().orc_fast9_16(_p(img), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(threshold), C.c_int(int(nonmax)),
                           _p(ox), _p(oy), _p(os_), C.c_int(cap))
    return ox[:n].copy(), oy[:n].copy(), os_[:n].copy()


def level_candidates(img, ini_th=20, min_th=7):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = w * h // 4 + 16
    cx = np.zeros(cap, 'f4'); cy = np.zeros(cap, 'f4'); cr = np.zeros(cap, 'f4')
    n = lib().orc_level_candidates(_p(img), C.c_int(w), C.c_int(w), C.c_int(h), C.c_int(ini_th), C.c_int(min_th),
                                   _p(cx), _p(cy), _p(cr), C.c_int(cap))
    return cx[:n].copy(), cy[:n].copy(), cr[:n].copy()


def distribute_octree(kx, ky, kr, minX, maxX, minY, maxY, N):
    kx = np.ascontiguousarray(kx, 'f4'); ky = np.ascontiguousarray(ky, 'f4'); kr = np.ascontiguousarray(kr, 'f4')
    cap = max(4 * N + 16, 16)
    out = np.zeros(cap, 'i4')
    n = lib().orc_distribute_octree(_p(kx), _p(ky), _p(kr), C.c_int(len(kx)), C.c_int(minX), C.c_int(maxX),
                                    C.c_int(minY), C.c_int(maxY), C.c_int(N), _p(out), C.c_int(cap))
    return out[:n].copy()


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def ic_angle(img, x, y, umax):
    img = np.ascontiguousarray(img, np.uint8)
    umax = np.ascontiguousarray(umax, 'i4')
    return float(lib().orc_ic_angle(_p(img), C.c_int(img.shape[1]), C.c_float(x), C.c_float(y), _p(umax)))


def descriptor(blur, x, y, angle):
    blur = np.ascontiguousarray(blur, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_descriptor(_p(blur), C.c_int(blur.shape[1]), C.c_float(x), C.c_float(y), C.c_float(angle), _p(d))
    return d


def orb_extract(gray, nfeatures=1000, scale=1.2, nlevels=8, ini_th=20, min_th=7, want_pyr=False):
    """ORBextractor::operator() restatement. Returns (keypoints[KP_DTYPE], desc[N,32] u8[, pyr, ncand])."""
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    cap = 4 * nfeatures + 64
    kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
    ncand = np.zeros(nlevels, 'i4')
    pyr = None
    if want_pyr:
        tot = sum(a * b for a, b in level_sizes(w, h, scale, nlevels))
        pyr = np.zeros(tot, np.uint8)
    n = lib().orc_orb_extract(_p(gray), C.c_int(w), C.c_int(h), C.c_int(w), C.c_int(nfeatures), C.c_float(scale),
                              C.c_int(nlevels), C.c_int(ini_th), C.c_int(min_th), _p(kps), _p(desc), C.c_int(cap),
                              _p(pyr) if want_pyr else None, _p(ncand))
    if n == -2:
        raise ValueError('a pyramid level is more than twice as tall as wide: DistributeOctTree is undefined there in the reference (nIni = 0)')
    assert 0 <= n <= cap
    if want_pyr:
        return kps[:n].copy(), desc[:n].copy(), pyr, ncand
    return kps[:n].copy(), desc[:n].copy()


# ---- matcher stage (oracle/match_oracle.c) ---------------------------------------------------
def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return int(lib().orc_descriptor_distance(_p(a), _p(b)))


def compute_stereo_from_rgbd(keys, depth_u16, bf=40.0, depth_factor=5000.0):
    """Tracking.cc:229-230 + Frame::ComputeStereoFromRGBD; returns (uright, zdepth)"""
    keys = np.ascontiguousarray(keys)
    raw = np.ascontiguousarray(depth_u16, np.uint16)
    h, w = raw.shape
    dep = np.zeros((h, w), 'f4')
    lib().orc_depth_convert(_p(raw), C.c_size_t(raw.size), C.c_float(np.float32(1.0) / np.float32(depth_factor)), _p(dep))
    n = len(keys)
    ur = np.zeros(n, 'f4'); z = np.zeros(n, 'f4')
    lib().orc_compute_stereo_from_rgbd(C.c_int(n), _p(keys), _p(keys), _p(dep), C.c_int(w), C.c_float(bf), _p(ur), _p(z))
    return ur, z


def unproject_stereo(keys, zdepth, Tcw, cam):
    keys = np.ascontiguousarray(keys); T = np.ascontiguousarray(Tcw, 'f4').reshape(16)
    n = len(keys)
    xw = np.zeros((n, 3), 'f4'); has = np.zeros(n, np.uint8)
    for i in range(n):
        o = np.zeros(3, 'f4')
        has[i] = lib().orc_unproject_stereo(_p(keys[i:i + 1]), C.c_float(zdepth[i]), _p(T), C.c_float(cam['fx']), C.c_float(cam['fy']),
                                            C.c_float(cam['cx']), C.c_float(cam['cy']), _p(o))
        xw[i] = o
    return xw, has


def search_by_projection_frame(cur, last, cam, scale_factors, th=15.0, mono=False, check_ori=True):
    """ORBmatcher::SearchByProjection(Cur, Last, th, bMono). cur/last: dicts of flattened frame arrays
    (keys, desc, uright, Tcw; last additionally has_mp, outlier, xw, obs, mpdesc). Returns (cur_match, nmatches)."""
    ck = np.ascontiguousarray(cur['keys']); cd = np.ascontiguousarray(cur['desc'], np.uint8); cu = np.ascontiguousarray(cur['uright'], 'f4')
    cT = np.ascontiguousarray(cur['Tcw'], 'f4').reshape(16)
    lk = np.ascontiguousarray(last['keys']); lh = np.ascontiguousarray(last['has_mp'], np.uint8); lo = np.ascontiguousarray(last['outlier'], np.uint8)
    lx = np.ascontiguousarray(last['xw'], 'f4'); lb = np.ascontiguousarray(last['obs'], 'i4'); lm = np.ascontiguousarray(last['mpdesc'], np.uint8)
    lT = np.ascontiguousarray(last['Tcw'], 'f4').reshape(16)
    sf = np.ascontiguousarray(scale_factors, 'f4')
    match = np.full(len(ck), -1, 'i4')
    L = lib()
    L.orc_search_by_projection_frame.restype = C.c_int
    n = L.orc_search_by_projection_frame(
        C.c_int(len(ck)), _p(ck), _p(cd), _p(cu), _p(cT),
        C.c_int(len(lk)), _p(lk), _p(lh), _p(lo), _p(lx), _p(lb), _p(lm), _p(lT),
        C.c_float(cam['fx']), C.c_float(cam['fy']), C.c_float(cam['cx']), C.c_float(cam['cy']), C.c_float(cam['bf']),
        C.c_float(cam.get('min_x', 0.0)), C.c_float(cam.get('max_x', 640.0)), C.c_float(cam.get('min_y', 0.0)), C.c_float(cam.get('max_y', 480.0)),
        _p(sf), C.c_float(th), C.c_int(int(mono)), C.c_int(int(check_ori)), _p(match))
    return match, int(n)


# ---- pose optimisation (oracle/poseopt_oracle.c) ---------------------------------------------
def pose_optimization(frame, cam, inv_level_sigma2, want_trace=False):
    """Optimizer::PoseOptimization restatement.  Returns (n_inliers, Tcw[4,4] f32, outlier[N][, trace, trace_n])."""
    k = np.ascontiguousarray(frame['keys']); ur = np.ascontiguousarray(frame['uright'], 'f4')
    has = np.ascontiguousarray(frame['has_mp'], np.uint8); xw = np.ascontiguousarray(frame['xw'], 'f4')
    T = np.ascontiguousarray(frame['Tcw'], 'f4').reshape(16).copy()
    is2 = np.ascontiguousarray(inv_level_sigma2, 'f4')
    n = len(k)
    out = np.zeros(max(n, 1), np.uint8)
    trace = np.zeros(4 * 11 * 3, 'f8'); tn = np.zeros(4, 'i4')
    L = lib(); L.orc_pose_optimization.restype = C.c_int
    r = L.orc_pose_optimization(C.c_int(n), _p(k), _p(ur), _p(is2), _p(has), _p(xw), C.c_float(cam['fx']), C.c_float(cam['fy']),
                                C.c_float(cam['cx']), C.c_float(cam['cy']), C.c_float(cam['bf']), _p(T), _p(out), _p(trace), _p(tn))
    if want_trace:
        return int(r), T.reshape(4, 4), out[:n], trace.reshape(4, 11, 3), tn
    return int(r), T.reshape(4, 4), out[:n]


# ---- local bundle adjustment (oracle/localba_oracle.c) -----------------------------------------
def local_ba(problem, cam, stop_flag=None):
    """Optimizer::LocalBundleAdjustment restatement on the flattened graph.  Returns (poses[np,4,4], points[nl,3], erase[ne], trace[2,15,3], iters[2])."""
    poses = np.ascontiguousarray(problem['poses'], 'f4').reshape(-1, 16).copy(); fixed = np.ascontiguousarray(problem['pose_fixed'], np.uint8)
    pts = np.ascontiguousarray(problem['points'], 'f4').copy()
    ep = np.ascontiguousarray(problem['edge_pose'], 'i4'); el = np.ascontiguousarray(problem['edge_point'], 'i4')
    eo = np.ascontiguousarray(problem['edge_obs'], 'f4'); ei = np.ascontiguousarray(problem['edge_info'], 'f4')
    erase = np.zeros(len(ep), np.uint8); trace = np.zeros(2 * 15 * 3, 'f8'); iters = np.zeros(2, 'i4')
    st = None if stop_flag is None else _p(np.ascontiguousarray(stop_flag, 'i4'))
    lib().orc_local_ba(C.c_int(len(poses)), _p(poses), _p(fixed), C.c_int(len(pts)), _p(pts), C.c_int(len(ep)), _p(ep), _p(el), _p(eo), _p(ei),
                       C.c_float(cam['fx']), C.c_float(cam['fy']), C.c_float(cam['cx']), C.c_float(cam['cy']), C.c_float(cam['bf']), st,
                       _p(erase), _p(trace), _p(iters))
    return poses.reshape(-1, 4, 4), pts, erase, trace.reshape(2, 15, 3), iters


def search_by_projection_local(cur, lm, cam, scale_factors, th=3.0, nnratio=0.8, viewing_cos_limit=0.5):
    """Tracking::SearchLocalPoints inner work: isInFrustum + ORBmatcher(0.8).SearchByProjection(F, vpMapPoints, th).
    cur: keys, desc, uright, Tcw [, mp_obs]; lm (local map): xw, normal, min_dist, max_dist, desc, obs, skip.
    Returns (cur_match[Nc], nmatches, in_view[Nm])."""
    ck = np.ascontiguousarray(cur['keys']); cd = np.ascontiguousarray(cur['desc'], np.uint8); cu = np.ascontiguousarray(cur['uright'], 'f4')
    cT = np.ascontiguousarray(cur['Tcw'], 'f4').reshape(16)
    co = np.ascontiguousarray(cur.get('mp_obs', np.full(len(ck), -1)), 'i4')
    xw = np.ascontiguousarray(lm['xw'], 'f4'); nr = np.ascontiguousarray(lm['normal'], 'f4'); mnd = np.ascontiguousarray(lm['min_dist'], 'f4'); mxd = np.ascontiguousarray(lm['max_dist'], 'f4')
    md = np.ascontiguousarray(lm['desc'], np.uint8); mo = np.ascontiguousarray(lm['obs'], 'i4'); ms = np.ascontiguousarray(lm['skip'], np.uint8)
    sf = np.ascontiguousarray(scale_factors, 'f4')
    match = np.full(max(len(ck), 1), -1, 'i4'); inview = np.zeros(max(len(xw), 1), np.uint8)
    L = lib(); L.orc_search_by_projection_local.restype = C.c_int
    n = L.orc_search_by_projection_local(
        C.c_int(len(ck)), _p(ck), _p(cd), _p(cu), _p(cT), _p(co),
        C.c_int(len(xw)), _p(xw), _p(nr), _p(mnd), _p(mxd), _p(md), _p(mo), _p(ms),
        C.c_float(cam['fx']), C.c_float(cam['fy']), C.c_float(cam['cx']), C.c_float(cam['cy']), C.c_float(cam['bf']),
        C.c_float(cam.get('min_x', 0.0)), C.c_float(cam.get('max_x', 640.0)), C.c_float(cam.get('min_y', 0.0)), C.c_float(cam.get('max_y', 480.0)),
        _p(sf), C.c_int(len(sf)), C.c_float(np.log(np.float32(sf[1]))), C.c_float(th), C.c_float(nnratio), C.c_float(viewing_cos_limit), _p(match), _p(inview))
    return match[:len(ck)], int(n), inview[:len(xw)]


# ---- mask inputs: pyramidal LK + RANSAC fundamental matrix (oracle/flow_oracle.c) ---------------
def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    dw, dh = (w + 1) // 2, (h + 1) // 2
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_pyrdown_u8(_p(img), C.c_int(w), C.c_int(h), C.c_int(w), _p(dst), C.c_int(dw), C.c_int(dh), C.c_int(dw))
    return dst


def scharr_deriv(img):
    img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
    d = np.zeros((h, w, 2), np.int16)
    lib().orc_scharr_deriv(_p(img), C.c_int(w), C.c_int(h), C.c_int(w), _p(d))
    return d


def lk_pyr(I, J, pts, win=21, max_level=3, max_count=30, epsilon=0.01, acc_mode=1, want_iters=False):
    """cv::calcOpticalFlowPyrLK(I, J, pts, ...) restatement: (next_pts[n,2] f32, status[n] u8 [, iters[n,levels]])."""
    I = np.ascontiguousarray(I, np.uint8); J = np.ascontiguousarray(J, np.uint8); h, w = I.shape
    pts = np.ascontiguousarray(pts, 'f4').reshape(-1, 2); n = len(pts)
    out = np.zeros((max(n, 1), 2), 'f4'); st = np.zeros(max(n, 1), np.uint8); it = np.zeros((max(n, 1), 8), 'i4')
    itl = np.zeros((max(n, 1) * 8,), 'i4')
    nl = lib().orc_lk_pyr(_p(I), _p(J), C.c_int(w), C.c_int(h), C.c_int(w), _p(pts), C.c_int(n), _p(out), _p(st), C.c_int(win), C.c_int(max_level),
                          C.c_int(max_count), C.c_double(epsilon), C.c_int(acc_mode), _p(itl))
    if want_iters:
        return out[:n], st[:n], itl[:n * nl].reshape(n, nl)
    return out[:n], st[:n]


def fm_select(cur, prev, pre_have_dynamic, boxes):
    cur = np.ascontiguousarray(cur, 'f4').reshape(-1, 2); prev = np.ascontiguousarray(prev, 'f4').reshape(-1, 2); n = len(cur)
    boxes = np.ascontiguousarray(boxes, 'f4').reshape(-1, 4)
    co = np.zeros((max(n, 1), 2), 'f4'); po = np.zeros((max(n, 1), 2), 'f4')
    m = lib().orc_fm_select(_p(cur), _p(prev), C.c_int(n), C.c_int(int(pre_have_dynamic)), _p(boxes), C.c_int(len(boxes)), _p(co), _p(po))
    return co[:m], po[:m]


def fm_run7point(m1, m2):
    m1 = np.ascontiguousarray(m1, 'f4').reshape(7, 2); m2 = np.ascontiguousarray(m2, 'f4').reshape(7, 2)
    f = np.zeros((3, 9), 'f8')
    n = lib().orc_fm_run7point(_p(m1), _p(m2), _p(f))
    return f[:max(n, 0)].reshape(-1, 3, 3)


def find_fundamental_ransac(m1, m2, threshold=1.0, confidence=0.99):
    """cv::findFundamentalMat(m1, m2, FM_RANSAC, threshold, confidence): (ok, F[3,3] f64, inlier mask, stats[iters, best_iter, best_root, inliers])."""
    m1 = np.ascontiguousarray(m1, 'f4').reshape(-1, 2); m2 = np.ascontiguousarray(m2, 'f4').reshape(-1, 2); n = len(m1)
    F = np.zeros(9, 'f8'); mask = np.zeros(max(n, 1), np.uint8); stats = np.zeros(4, 'i4')
    ok = lib().orc_find_fundamental_ransac(_p(m1), _p(m2), C.c_int(n), C.c_double(threshold), C.c_double(confidence), _p(F), _p(mask), _p(stats))
    return int(ok), F.reshape(3, 3), mask[:n], stats


def solve_cubic(c):
    c = np.ascontiguousarray(c, 'f8'); r = np.zeros(3, 'f8')
    n = lib().orc_solve_cubic(_p(c), _p(r))
    return int(n), r


def dynamic_mask(cur, prev, F, boxes, have_dynamic, nfeatures=1000):
    """Frame.cc:556-604 keep flags + restore rule (C twin of detector_oracle.dynamic_mask): (keep[n] bool, restored)"""
    cur = np.ascontiguousarray(cur, 'f4').reshape(-1, 2); prev = np.ascontiguousarray(prev, 'f4').reshape(-1, 2)
    F = np.ascontiguousarray(F, 'f8').reshape(9); boxes = np.ascontiguousarray(boxes, 'f4').reshape(-1, 4)
    keep = np.zeros(max(len(cur), 1), np.uint8)
    r = lib().orc_dynamic_mask(_p(cur), _p(prev), C.c_int(len(cur)), _p(F), _p(boxes), C.c_int(len(boxes)), C.c_int(int(bool(have_dynamic))), C.c_int(nfeatures), _p(keep))
    return keep[:len(cur)].astype(bool), bool(r)


def bundle_adjustment(problem, cam, n_iterations=5, robust=True, stop_flag=None):
    """Optimizer::BundleAdjustment restatement on the flattened graph.  Returns (poses[np,4,4], points[nl,3], trace[n_iterations,3], iterations)."""
    poses = np.ascontiguousarray(problem['poses'], 'f4').reshape(-1, 16).copy(); fixed = np.ascontiguousarray(problem['pose_fixed'], np.uint8)
    pts = np.ascontiguousarray(problem['points'], 'f4').copy()
    ep = np.ascontiguousarray(problem['edge_pose'], 'i4'); el = np.ascontiguousarray(problem['edge_point'], 'i4')
    eo = np.ascontiguousarray(problem['edge_obs'], 'f4'); ei = np.ascontiguousarray(problem['edge_info'], 'f4')
    trace = np.zeros(max(n_iterations, 1) * 3, 'f8'); iters = np.zeros(1, 'i4')
    st = None if stop_flag is None else _p(np.ascontiguousarray(stop_flag, 'i4'))
    lib().orc_bundle_adjustment(C.c_int(len(poses)), _p(poses), _p(fixed), C.c_int(len(pts)), _p(pts), C.c_int(len(ep)), _p(ep), _p(el), _p(eo), _p(ei),
                                C.c_float(cam['fx']), C.c_float(cam['fy']), C.c_float(cam['cx']), C.c_float(cam['cy']), C.c_float(cam['bf']),
                                C.c_int(n_iterations), C.c_int(int(bool(robust))), st, _p(trace), _p(iters))
    return poses.reshape(-1, 4, 4), pts, trace.reshape(-1, 3), int(iters[0])


# ---- LocalMapping gates (oracle/match_oracle.c) ----------------------------------------------------
def hamming_matrix(a, b):
    a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
    out = np.zeros((len(a), len(b)), np.uint16)
    lib().orc_hamming_matrix(_p(a), C.c_int(len(a)), _p(b), C.c_int(len(b)), _p(out))
    return out


def search_for_triangulation(kf1, kf2, F12, only_stereo, cam2, scale2, sigma2_2, check_ori=True):
    """ORBmatcher::SearchForTriangulation restatement: (nmatches, pairs[n,2])"""
    k1 = np.ascontiguousarray(kf1['keys']); d1 = np.ascontiguousarray(kf1['desc'], np.uint8); u1 = np.ascontiguousarray(kf1['uright'], 'f4')
    h1 = np.ascontiguousarray(kf1['has_mp'], np.uint8); n1 = np.ascontiguousarray(kf1['feat_node'], 'i4'); c1 = np.ascontiguousarray(kf1['cam_center'], 'f4')
    k2 = np.ascontiguousarray(kf2['keys']); d2 = np.ascontiguousarray(kf2['desc'], np.uint8); u2 = np.ascontiguousarray(kf2['uright'], 'f4')
    h2 = np.ascontiguousarray(kf2['has_mp'], np.uint8); n2 = np.ascontiguousarray(kf2['feat_node'], 'i4'); T2 = np.ascontiguousarray(kf2['Tcw'], 'f4').reshape(16)
    F = np.ascontiguousarray(F12, 'f4').reshape(9); sf = np.ascontiguousarray(scale2, 'f4'); sg = np.ascontiguousarray(sigma2_2, 'f4')
    pairs = np.zeros((max(len(k1), 1), 2), 'i4')
    L = lib(); L.orc_search_for_triangulation.restype = C.c_int
    n = L.orc_search_for_triangulation(C.c_int(len(k1)), _p(k1), _p(d1), _p(u1), _p(h1), _p(n1), _p(c1), C.c_int(len(k2)), _p(k2), _p(d2), _p(u2), _p(h2), _p(n2), _p(T2),
                                       _p(F), C.c_float(cam2['fx']), C.c_float(cam2['fy']), C.c_float(cam2['cx']), C.c_float(cam2['cy']), _p(sf), _p(sg),
                                       C.c_int(int(bool(only_stereo))), C.c_int(int(bool(check_ori))), _p(pairs))
    return int(n), pairs[:n].copy()


def search_by_bow(kf, F, nnratio=0.7, check_ori=True):
    kk = np.ascontiguousarray(kf['keys']); dk = np.ascontiguousarray(kf['desc'], np.uint8); gk = np.ascontiguousarray(kf['good_mp'], np.uint8); nk = np.ascontiguousarray(kf['feat_node'], 'i4')
    kq = np.ascontiguousarray(F['keys']); dq = np.ascontiguousarray(F['desc'], np.uint8); nq = np.ascontiguousarray(F['feat_node'], 'i4')
    match = np.full(max(len(kq), 1), -1, 'i4')
    L = lib(); L.orc_search_by_bow.restype = C.c_int
    n = L.orc_search_by_bow(C.c_int(len(kk)), _p(kk), _p(dk), _p(gk), _p(nk), C.c_int(len(kq)), _p(kq), _p(dq), _p(nq), C.c_float(nnratio), C.c_int(int(bool(check_ori))), _p(match))
    return int(n), match[:len(kq)].copy()


def search_for_initialization(F1, F2, prev_matched, window, cam, nnratio=0.9, check_ori=True):
    a = np.ascontiguousarray(F1['keys']); da = np.ascontiguousarray(F1['desc'], np.uint8); b = np.ascontiguousarray(F2['keys']); db = np.ascontiguousarray(F2['desc'], np.uint8)
    pm = np.ascontiguousarray(prev_matched, 'f4').reshape(-1, 2).copy(); m = np.full(max(len(a), 1), -1, 'i4')
    L = lib(); L.orc_search_for_initialization.restype = C.c_int
    n = L.orc_search_for_initialization(C.c_int(len(a)), _p(a), _p(da), C.c_int(len(b)), _p(b), _p(db), _p(pm), C.c_int(int(window)), C.c_float(nnratio), C.c_int(int(bool(check_ori))),
                                        C.c_float(cam.get('min_x', 0.0)), C.c_float(cam.get('max_x', 640.0)), C.c_float(cam.get('min_y', 0.0)), C.c_float(cam.get('max_y', 480.0)), _p(m))
    return int(n), m[:len(a)].copy(), pm


def search_by_bow_kf(kf1, kf2, nnratio=0.75, check_ori=True):
    a = [np.ascontiguousarray(kf1['keys']), np.ascontiguousarray(kf1['desc'], np.uint8), np.ascontiguousarray(kf1['good_mp'], np.uint8), np.ascontiguousarray(kf1['feat_node'], 'i4')]
    b = [np.ascontiguousarray(kf2['keys']), np.ascontiguousarray(kf2['desc'], np.uint8), np.ascontiguousarray(kf2['good_mp'], np.uint8), np.ascontiguousarray(kf2['feat_node'], 'i4')]
    match = np.full(max(len(a[0]), 1), -1, 'i4')
    L = lib(); L.orc_search_by_bow_kf.restype = C.c_int
    n = L.orc_search_by_bow_kf(C.c_int(len(a[0])), *[_p(x) for x in a], C.c_int(len(b[0])), *[_p(x) for x in b], C.c_float(nnratio), C.c_int(int(bool(check_ori))), _p(match))
    return int(n), match[:len(a[0])].copy()


def fuse_search(kf, m, cam, scale_factors, inv_level_sigma2, th=3.0):
    k = np.ascontiguousarray(kf['keys']); d = np.ascontiguousarray(kf['desc'], np.uint8); u = np.ascontiguousarray(kf['uright'], 'f4'); T = np.ascontiguousarray(kf['Tcw'], 'f4').reshape(16)
    xw = np.ascontiguousarray(m['xw'], 'f4'); nr = np.ascontiguousarray(m['normal'], 'f4'); mn = np.ascontiguousarray(m['min_dist'], 'f4'); mx = np.ascontiguousarray(m['max_dist'], 'f4')
    md = np.ascontiguousarray(m['desc'], np.uint8); sk = np.ascontiguousarray(m['skip'], np.uint8)
    sf = np.ascontiguousarray(scale_factors, 'f4'); is2 = np.ascontiguousarray(inv_level_sigma2, 'f4')
    bi = np.full(max(len(xw), 1), -1, 'i4'); bd = np.full(max(len(xw), 1), 256, 'i4')
    L = lib(); L.orc_fuse_search.restype = C.c_int
    n = L.orc_fuse_search(C.c_int(len(k)), _p(k), _p(d), _p(u), _p(T), C.c_int(len(xw)), _p(xw), _p(nr), _p(mn), _p(mx), _p(md), _p(sk),
                          C.c_float(cam['fx']), C.c_float(cam['fy']), C.c_float(cam['cx']), C.c_float(cam['cy']), C.c_float(cam['bf']),
                          C.c_float(cam.get('min_x', 0.0)), C.c_float(cam.get('max_x', 640.0)), C.c_float(cam.get('min_y', 0.0)), C.c_float(cam.get('max_y', 480.0)),
                          _p(sf), _p(is2), C.c_int(len(sf)), C.c_float(np.log(np.float32(sf[1]))), C.c_float(th), _p(bi), _p(bd))
    return int(n), bi[:len(xw)].copy(), bd[:len(xw)].copy()


def search_by_projection_kf(F, kf, cam, scale_factors, th, orb_dist, check_ori=True):
    ck = np.ascontiguousarray(F['keys']); cd = np.ascontiguousarray(F['desc'], np.uint8); ch = np.ascontiguousarray(F['has_mp'], np.uint8); T = np.ascontiguousarray(F['Tcw'], 'f4').reshape(16)
    kk = np.ascontiguousarray(kf['keys']); ok = np.ascontiguousarray(kf['ok'], np.uint8); xw = np.ascontiguousarray(kf['xw'], 'f4')
    mn = np.ascontiguousarray(kf['min_dist'], 'f4'); mx = np.ascontiguousarray(kf['max_dist'], 'f4'); md = np.ascontiguousarray(kf['desc'], np.uint8)
    sf = np.ascontiguousarray(scale_factors, 'f4')
    match = np.full(max(len(ck), 1), -1, 'i4')
    L = lib(); L.orc_search_by_projection_kf.restype = C.c_int
    n = L.orc_search_by_projection_kf(C.c_int(len(ck)), _p(ck), _p(cd), _p(ch), _p(T), C.c_int(len(kk)), _p(kk), _p(ok), _p(xw), _p(mn), _p(mx), _p(md),
                                      C.c_float(cam['fx']), C.c_float(cam['fy']), C.c_float(cam['cx']), C.c_float(cam['cy']),
                                      C.c_float(cam.get('min_x', 0.0)), C.c_float(cam.get('max_x', 640.0)), C.c_float(cam.get('min_y', 0.0)), C.c_float(cam.get('max_y', 480.0)),
                                      _p(sf), C.c_int(len(sf)), C.c_float(np.log(np.float32(sf[1]))), C.c_float(th), C.c_int(orb_dist), C.c_int(int(bool(check_ori))), _p(match))
    return int(n), match[:len(ck)].copy()


def _cam_args(cam):
    return [C.c_float(cam['fx']), C.c_float(cam['fy']), C.c_float(cam['cx']), C.c_float(cam['cy']),
            C.c_float(cam.get('min_x', 0.0)), C.c_float(cam.get('max_x', 640.0)), C.c_float(cam.get('min_y', 0.0)), C.c_float(cam.get('max_y', 480.0))]


def _points(m):
    return [np.ascontiguousarray(m['xw'], 'f4'), np.ascontiguousarray(m['normal'], 'f4'), np.ascontiguousarray(m['min_dist'], 'f4'), np.ascontiguousarray(m['max_dist'], 'f4'),
            np.ascontiguousarray(m['desc'], np.uint8), np.ascontiguousarray(m['skip'], np.uint8)]


def fuse_search_sim3(kf, Scw, m, cam, scale_factors, th):
    k = np.ascontiguousarray(kf['keys']); d = np.ascontiguousarray(kf['desc'], np.uint8); S = np.ascontiguousarray(Scw, 'f4').reshape(16)
    pts = _points(m); nm = len(pts[0]); sf = np.ascontiguousarray(scale_factors, 'f4')
    bi = np.full(max(nm, 1), -1, 'i4'); bd = np.full(max(nm, 1), 256, 'i4')
    L = lib(); L.orc_fuse_search_sim3.restype = C.c_int
    n = L.orc_fuse_search_sim3(C.c_int(len(k)), _p(k), _p(d), _p(S), C.c_int(nm), *[_p(x) for x in pts], *_cam_args(cam), _p(sf), C.c_int(len(sf)),
                               C.c_float(np.log(np.float32(sf[1]))), C.c_float(th), _p(bi), _p(bd))
    return int(n), bi[:nm].copy(), bd[:nm].copy()


def search_by_projection_sim3(kf, Scw, m, cam, scale_factors, th):
    k = np.ascontiguousarray(kf['keys']); d = np.ascontiguousarray(kf['desc'], np.uint8); mi = np.ascontiguousarray(kf['matched'], np.uint8); S = np.ascontiguousarray(Scw, 'f4').reshape(16)
    pts = _points(m); nm = len(pts[0]); sf = np.ascontiguousarray(scale_factors, 'f4')
    mo = np.full(max(len(k), 1), -1, 'i4')
    L = lib(); L.orc_search_by_projection_sim3.restype = C.c_int
    n = L.orc_search_by_projection_sim3(C.c_int(len(k)), _p(k), _p(d), _p(mi), _p(S), C.c_int(nm), *[_p(x) for x in pts], *_cam_args(cam), _p(sf), C.c_int(len(sf)),
                                        C.c_float(np.log(np.float32(sf[1]))), C.c_int(int(th)), _p(mo))
    return int(n), mo[:len(k)].copy()


def search_by_sim3(kf1, kf2, match12, s12, R12, t12, th, cam, scale_factors):
    def flat(kf):
        return [np.ascontiguousarray(kf['keys']), np.ascontiguousarray(kf['desc'], np.uint8), np.ascontiguousarray(kf['Tcw'], 'f4').reshape(16), np.ascontiguousarray(kf['mp_ok'], np.uint8),
                np.ascontiguousarray(kf['xw'], 'f4'), np.ascontiguousarray(kf['min_dist'], 'f4'), np.ascontiguousarray(kf['max_dist'], 'f4'), np.ascontiguousarray(kf['mp_desc'], np.uint8)]
    a = flat(kf1); b = flat(kf2); sf = np.ascontiguousarray(scale_factors, 'f4')
    R = np.ascontiguousarray(R12, 'f4').reshape(9); t = np.ascontiguousarray(t12, 'f4').reshape(3)
    m = np.full(max(len(a[0]), 1), -1, 'i4'); m[:len(a[0])] = np.asarray(match12, 'i4')
    L = lib(); L.orc_search_by_sim3.restype = C.c_int
    n = L.orc_search_by_sim3(C.c_int(len(a[0])), *[_p(x) for x in a], C.c_int(len(b[0])), *[_p(x) for x in b], *_cam_args(cam), _p(sf), C.c_int(len(sf)),
                             C.c_float(np.log(np.float32(sf[1]))), C.c_float(s12), _p(R), _p(t), C.c_float(th), _p(m))
    return int(n), m[:len(a[0])].copy()


def optimize_sim3(p1c, p2c, obs1, obs2, info1, info2, K1, K2, S12, th2=10.0, fix_scale=False):
    """Optimizer::OptimizeSim3 on flattened correspondences: (nIn, S12[8] = (qx,qy,qz,qw,tx,ty,tz,s), inlier[n], iterations[2])."""
    p1c = np.ascontiguousarray(p1c, 'f4').reshape(-1, 3); p2c = np.ascontiguousarray(p2c, 'f4').reshape(-1, 3); n = len(p1c)
    o1 = np.ascontiguousarray(obs1, 'f4').reshape(-1, 2); o2 = np.ascontiguousarray(obs2, 'f4').reshape(-1, 2)
    i1 = np.ascontiguousarray(info1, 'f4'); i2 = np.ascontiguousarray(info2, 'f4')
    k1 = np.ascontiguousarray(K1, 'f4'); k2 = np.ascontiguousarray(K2, 'f4')
    S = np.ascontiguousarray(S12, 'f8').copy(); inl = np.zeros(max(n, 1), np.uint8); it = np.zeros(2, 'i4')
    L = lib(); L.orc_optimize_sim3.restype = C.c_int
    nin = L.orc_optimize_sim3(C.c_int(n), _p(p1c), _p(p2c), _p(o1), _p(o2), _p(i1), _p(i2), _p(k1), _p(k2), _p(S), C.c_float(th2), C.c_int(int(bool(fix_scale))), _p(inl), _p(it))
    return int(nin), S, inl[:n].copy(), it


def optimize_essential_graph(S, fixed, e_i, e_j, e_meas, fix_scale=True, iterations=20):
    """the optimisation of Optimizer::OptimizeEssentialGraph on a flattened pose graph: (S_out[nv,8], stats[iterations, chi2 before, chi2 after])"""
    S = np.ascontiguousarray(S, 'f8').reshape(-1, 8).copy(); fx = np.ascontiguousarray(fixed, np.uint8)
    ei = np.ascontiguousarray(e_i, 'i4'); ej = np.ascontiguousarray(e_j, 'i4'); em = np.ascontiguousarray(e_meas, 'f8').reshape(-1, 8)
    st = np.zeros(3, 'f8')
    L = lib(); L.orc_optimize_essential_graph.restype = C.c_int
    L.orc_optimize_essential_graph(C.c_int(len(S)), _p(S), _p(fx), C.c_int(len(ei)), _p(ei), _p(ej), _p(em), C.c_int(int(bool(fix_scale))), C.c_int(iterations), _p(st))
    return S, st


def correct_map_points(xw, ref, Srw, corrected_Swr):
    xw = np.ascontiguousarray(xw, 'f4').reshape(-1, 3); ref = np.ascontiguousarray(ref, 'i4')
    a = np.ascontiguousarray(Srw, 'f8').reshape(-1, 8); c = np.ascontiguousarray(corrected_Swr, 'f8').reshape(-1, 8)
    out = np.zeros_like(xw)
    lib().orc_correct_map_points(C.c_int(len(xw)), _p(xw), _p(ref), _p(a), _p(c), _p(out))
    return out


# ---- DBoW2 vocabulary transform (bow_oracle.c) -------------------------------------------------------------------------------------------------------------------
class Vocabulary:
    """oracle-side ORBVocabulary: create from flat arrays or load a text / binary vocabulary file"""

    def __init__(self, k=None, L=None, parent=None, desc=None, weight=None, is_leaf=None, scoring=0, weighting=0, path=None):
        Lb = lib(); Lb.orc_voc_create.restype = C.c_void_p; Lb.orc_voc_load.restype = C.c_void_p
        if path is not None:
            self.h = Lb.orc_voc_load(path.encode())
            if not self.h: raise IOError(path)
        else:
            parent = np.ascontiguousarray(parent, 'i4'); desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, 'f8'); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            self.h = Lb.orc_voc_create(C.c_int(k), C.c_int(L), C.c_int(scoring), C.c_int(weighting), C.c_int(len(parent)), _p(parent), _p(desc), _p(weight), _p(is_leaf))
        info = np.zeros(6, 'i4'); Lb.orc_voc_info(C.c_void_p(self.h), _p(info))
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = (int(x) for x in info)

    def transform(self, descriptors, levelsup=4):
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32); n = len(d)
        word = np.full(max(n, 1), -1, 'i4'); w = np.zeros(max(n, 1), 'f8'); fn = np.full(max(n, 1), -1, 'i4')
        Lb = lib(); Lb.orc_voc_transform_features(C.c_void_p(self.h), C.c_int(n), _p(d), C.c_int(levelsup), _p(word), _p(w), _p(fn))
        ids = np.zeros(max(n, 1), 'i4'); bw = np.zeros(max(n, 1), 'f8')
        Lb.orc_voc_bow_vector.restype = C.c_int
        m = Lb.orc_voc_bow_vector(C.c_void_p(self.h), C.c_int(n), _p(word), _p(w), _p(ids), _p(bw))
        return ids[:m].copy(), bw[:m].copy(), fn[:n].copy(), word[:n].copy()

    def close(self):
        if self.h: lib().orc_voc_destroy(C.c_void_p(self.h)); self.h = None


def bow_score_l1(v1, v2):
    i1 = np.ascontiguousarray(v1[0], 'i4'); w1 = np.ascontiguousarray(v1[1], 'f8'); i2 = np.ascontiguousarray(v2[0], 'i4'); w2 = np.ascontiguousarray(v2[1], 'f8')
    Lb = lib(); Lb.orc_bow_score_l1.restype = C.c_double
    return float(Lb.orc_bow_score_l1(C.c_int(len(i1)), _p(i1), _p(w1), C.c_int(len(i2)), _p(i2), _p(w2)))


# ---- Sim3Solver (sim3solver_oracle.c) ---------------------------------------------------------------------------------------------------------------------------------
class Sim3SolverOracle:
    def __init__(self, x3dc1, x3dc2, max_err1, max_err2, K1, K2, fix_scale=True):
        a = np.ascontiguousarray(x3dc1, 'f4').reshape(-1, 3); b = np.ascontiguousarray(x3dc2, 'f4').reshape(-1, 3)
        e1 = np.ascontiguousarray(max_err1, 'f4'); e2 = np.ascontiguousarray(max_err2, 'f4'); k1 = np.ascontiguousarray(K1, 'f4'); k2 = np.ascontiguousarray(K2, 'f4')
        L = lib(); L.orc_s3_create.restype = C.c_void_p
        self.N = len(a); self.h = L.orc_s3_create(C.c_int(self.N), _p(a), _p(b), _p(e1), _p(e2), _p(k1), _p(k2), C.c_int(int(bool(fix_scale))))

    def set_ransac_parameters(self, probability=0.99, min_inliers=6, max_iterations=300):
        lib().orc_s3_set_ransac_parameters(C.c_void_p(self.h), C.c_double(probability), C.c_int(min_inliers), C.c_int(max_iterations))

    def max_iterations(self):
        L = lib(); L.orc_s3_max_iterations.restype = C.c_int; return int(L.orc_s3_max_iterations(C.c_void_p(self.h)))

    def iterate(self, n_iterations, rand_draws):
        d = np.ascontiguousarray(rand_draws, 'i4'); T = np.zeros(16, 'f4'); nm = C.c_int(); inl = np.zeros(max(self.N, 1), np.uint8); ni = C.c_int(); run = C.c_int()
        L = lib(); L.orc_s3_iterate.restype = C.c_int
        f = L.orc_s3_iterate(C.c_void_p(self.h), C.c_int(n_iterations), _p(d), _p(T), C.byref(nm), _p(inl), C.byref(ni), C.byref(run))
        return (T.reshape(4, 4) if f else None), bool(nm.value), inl[:self.N].astype(bool), int(ni.value), int(run.value)

    def estimate(self):
        R = np.zeros(9, 'f4'); t = np.zeros(3, 'f4'); s = C.c_float()
        lib().orc_s3_best(C.c_void_p(self.h), _p(R), _p(t), C.byref(s)); return R.reshape(3, 3), t, float(s.value)

    def close(self):
        if self.h: lib().orc_s3_destroy(C.c_void_p(self.h)); self.h = None


def glibc_rand_sequence(seed, n):
    out = np.zeros(max(n, 1), 'i4'); lib().orc_glibc_rand_sequence(C.c_uint(seed), C.c_int(n), _p(out)); return out[:n]


def sim3_horn(P1, P2, fix_scale=False):
    """ComputeSim3 on one triple: P1, P2 = 3 x 3 (column i = point i) -> (R, t, s)"""
    a = np.ascontiguousarray(P1, 'f4').reshape(9); b = np.ascontiguousarray(P2, 'f4').reshape(9); R = np.zeros(9, 'f4'); t = np.zeros(3, 'f4'); s = C.c_float()
    lib().orc_s3_compute(_p(a), _p(b), C.c_int(int(bool(fix_scale))), _p(R), _p(t), C.byref(s)); return R.reshape(3, 3), t, float(s.value)


def update_normal_and_depth(xw, obs_start, obs_center, ref_center, ref_level, scale_factors, normal, min_dist, max_dist):
    xw = np.ascontiguousarray(xw, 'f4').reshape(-1, 3); n = len(xw)
    st = np.ascontiguousarray(obs_start, 'i4'); oc = np.ascontiguousarray(obs_center, 'f4').reshape(-1, 3); rc = np.ascontiguousarray(ref_center, 'f4').reshape(-1, 3)
    rl = np.ascontiguousarray(ref_level, 'i4'); sf = np.ascontiguousarray(scale_factors, 'f4')
    nr = np.ascontiguousarray(normal, 'f4').reshape(-1, 3).copy(); mn = np.ascontiguousarray(min_dist, 'f4').copy(); mx = np.ascontiguousarray(max_dist, 'f4').copy()
    lib().orc_update_normal_and_depth(C.c_int(n), _p(xw), _p(st), _p(oc), _p(rc), _p(rl), _p(sf), C.c_int(len(sf)), _p(nr), _p(mn), _p(mx))
    return nr, mn, mx


def distinctive_descriptors(obs_start, obs_desc):
    st = np.ascontiguousarray(obs_start, 'i4'); n = len(st) - 1; d = np.ascontiguousarray(obs_desc, np.uint8).reshape(-1, 32)
    best = np.full(max(n, 1), -1, 'i4')
    lib().orc_distinctive_descriptors(C.c_int(n), _p(st), _p(d), _p(best))
    return best[:n].copy()


def triangulate_pairs(pairs, kf1, kf2, cam, scale_factors, level_sigma2):
    pr = np.ascontiguousarray(pairs, 'i4').reshape(-1, 2); npairs = len(pr)
    def flat(k):
        return (np.ascontiguousarray(k['keys_un']), np.ascontiguousarray(k.get('keys', k['keys_un'])), np.ascontiguousarray(k['uright'], 'f4'), np.ascontiguousarray(k['depth'], 'f4'),
                np.ascontiguousarray(k['Tcw'], 'f4').reshape(16))
    a = flat(kf1); b = flat(kf2); sf = np.ascontiguousarray(scale_factors, 'f4'); sg = np.ascontiguousarray(level_sigma2, 'f4')
    ok = np.zeros(max(npairs, 1), np.uint8); x = np.zeros((max(npairs, 1), 3), 'f4')
    L = lib(); L.orc_triangulate_pairs.restype = C.c_int
    n = L.orc_triangulate_pairs(C.c_int(npairs), _p(pr), *[_p(v) for v in a], *[_p(v) for v in b], C.c_float(cam['fx']), C.c_float(cam['fy']), C.c_float(cam['cx']), C.c_float(cam['cy']),
                                C.c_float(cam['bf']), _p(sf), _p(sg), C.c_float(sf[1]), _p(ok), _p(x))
    return int(n), ok[:npairs].astype(bool), x[:npairs].copy()
