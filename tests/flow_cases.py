"""Shared assertions for the mask-input stage (LK flow + RANSAC F): run against the kernel-logic emulator (CPU tier) and the device (-m gpu)."""
import numpy as np
from sg_slam_amd import synth
from sg_slam_amd.flow import OpticalFlowLK, find_fundamental_mat, fundamental_ransac_batch_dev
from sg_slam_amd.capi import KP_DTYPE


def frame_pair(t=11, seed=1234):
    S = synth.PlaneStream(seed=seed)
    return S, S.frame(t - 1)[0], S.frame(t)[0]


def two_view(n=400, seed=0, outlier_every=4, noise=0.3):
    """Synthetic two-view correspondences (3-D points, TUM3 intrinsics), float32 pixels, gross outliers on every `outlier_every`-th pair."""
    rng = np.random.RandomState(seed)
    K = np.array([[535.4, 0, 320.1], [0, 539.2, 247.6], [0, 0, 1.]])
    w = rng.normal(0, 0.03, 3); th = np.linalg.norm(w); k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(0, 0.08, 3)
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 6, n)]
    x1 = (K @ X.T).T; x1 = x1[:, :2] / x1[:, 2:]
    X2 = (R @ X.T).T + t; x2 = (K @ X2.T).T; x2 = x2[:, :2] / x2[:, 2:]
    x2 = x2 + rng.normal(0, noise, x2.shape)
    if outlier_every:
        x2[::outlier_every] += rng.uniform(-40, 40, x2[::outlier_every].shape)
    return x1.astype('f4'), x2.astype('f4')


def check_pyramid(lib, orc):
    """pyrDown chain, byte for byte (incl. an odd-sized image: reflect paths, padded quads)"""
    for (w, h, seed) in ((640, 480, 1), (173, 131, 2)):
        rng = np.random.RandomState(seed)
        img = rng.randint(0, 256, (h, w)).astype(np.uint8)
        img[h // 4:h // 2, w // 4:w // 2] = 200
        fl = OpticalFlowLK(width=w, height=h, lib=lib)
        pts = np.array([[w / 2, h / 2]], 'f4')
        fl(img, img, pts)                                   # builds slot 0 (from-image) and slot 1
        ref = img
        for l in range(fl.levels):
            got = fl.debug_level(0, 0, l)
            assert got.shape == ref.shape
            assert (got == ref).all(), f'pyramid level {l} differs ({w}x{h})'
            ref = orc.pyr_down(ref)
        fl.close()


def check_lk_pair(lib, orc, t=11, n_extra=40):
    """tracked positions bit-identical to the oracle's exact-accumulation variant; within 0.01 px of its float-accumulation variant"""
    S, prev, cur = frame_pair(t)
    k, _ = orc.orb_extract(cur)
    pts = np.stack([k['x'], k['y']], 1).astype('f4')
    rng = np.random.RandomState(3)
    extra = np.c_[rng.uniform(-30, 670, n_extra), rng.uniform(-30, 510, n_extra)].astype('f4')      # border / outside points: skip + status rules
    pts = np.concatenate([pts, extra, np.array([[0.5, 0.5], [639.4, 479.4], [5.25, 474.75]], 'f4')])
    fl = OpticalFlowLK(lib=lib)
    got, st = fl(cur, prev, pts)
    ref, rst = orc.lk_pyr(cur, prev, pts, acc_mode=1)
    assert (st == rst).all()
    assert (got.view(np.uint32) == ref.view(np.uint32)).all(), f'LK positions differ from the oracle: max {np.abs(got - ref).max()}'
    ref0, _ = orc.lk_pyr(cur, prev, pts, acc_mode=0)
    ok = st > 0
    assert np.abs(got[ok] - ref0[ok]).max() < 0.01           # float-order noise of OpenCV's x86 accumulation stays far below the 0.2 / 1.0 px mask thresholds
    A = synth.flow_affine(S, t, t - 1)
    nk = len(k)
    gt = pts[:nk] @ A[:, :2].T + A[:, 2]
    assert np.median(np.linalg.norm(got[:nk] - gt, axis=1)) < 0.15
    fl.close()
    return got


def check_lk_textureless(lib, orc):
    """flat image: minimum-eigenvalue rule leaves every point where the pyramid guess put it, status 0"""
    img = np.full((480, 640), 90, np.uint8)
    pts = np.array([[100.5, 100.25], [320, 240], [600.75, 20.5]], 'f4')
    fl = OpticalFlowLK(lib=lib)
    got, st = fl(img, img, pts)
    ref, rst = orc.lk_pyr(img, img, pts)
    assert (st == rst).all() and (st == 0).all()
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
    fl.close()


def check_lk_stream(lib, orc, xp, S=2, T=4):
    """streaming form on a batch: first call only builds the pyramid; later calls track into the previous call's frames"""
    gen = synth.PlaneStream(seed=1234)
    fl = OpticalFlowLK(max_batch=S, lib=lib)
    cap = 1100
    for t in range(T):
        frames = np.stack([gen.frame(5 + 40 * s + t)[0] for s in range(S)])
        keys = np.zeros((S, cap), KP_DTYPE); n = np.zeros(S, 'i4')
        for s in range(S):
            k, _ = orc.orb_extract(frames[s]); n[s] = len(k); keys[s, :len(k)] = k
        d_fr, d_k, d_n = xp(frames), xp(keys.view(np.uint8).reshape(S, cap, 28)), xp(n)
        d_out = xp(np.zeros((S, cap, 2), 'f4')); d_st = xp(np.zeros((S, cap), np.uint8))
        have = fl.lk_batch_dev(d_fr, 640, S, d_k, d_n, cap, d_out, d_st)
        assert have == (t > 0)
        if t > 0:
            out = np.asarray(d_out.cpu() if hasattr(d_out, 'cpu') else d_out); stt = np.asarray(d_st.cpu() if hasattr(d_st, 'cpu') else d_st)
            for s in range(S):
                pts = np.stack([keys[s, :n[s]]['x'], keys[s, :n[s]]['y']], 1)
                ref, rst = orc.lk_pyr(frames[s], prev_frames[s], pts)
                assert (out[s, :n[s]].view(np.uint32) == ref.view(np.uint32)).all() and (stt[s, :n[s]] == rst).all()
        prev_frames = frames
    fl.close()


def check_ransac_host(lib, orc, exact_lmeds=True):
    """findFundamentalMat: same RANSAC trajectory (iterations, winning sample / root, inlier count) and F within 1e-9 of the oracle"""
    for seed, (n, oe) in enumerate(((400, 4), (1000, 3), (60, 2), (15, 0), (500, 0))):
        x1, x2 = two_view(n, seed, oe)
        ok, F, st = find_fundamental_mat(x1, x2, lib=lib)
        rok, rF, rmask, rst = orc.find_fundamental_ransac(x1, x2)
        assert ok == rok == 1
        assert (st == rst).all(), (st, rst)
        assert np.abs(F - rF).max() <= 1e-9 * np.abs(rF).max()
        # the property the mask needs: static (inlier) pairs sit on their epipolar lines
        l = (F @ np.c_[x1, np.ones(n)].T).T
        d = np.abs((l * np.c_[x2, np.ones(n)]).sum(1)) / np.hypot(l[:, 0], l[:, 1])
        assert np.median(d[rmask > 0]) < 0.6
    lmeds_ok = 0
    for n in (0, 5, 7, 8, 9, 10, 12, 14):                    # < 7: empty; 7: 7-point directly; 8..14: OpenCV's LMedS branch
        for seed, every in ((9, 0), (10, 5), (11, 3)):              # no outliers / every fifth / every third pair an outlier
            x1, x2 = two_view(max(n, 1), seed, every)
            ok, F, st = find_fundamental_mat(x1[:n], x2[:n], lib=lib)
            rok, rF, rmask, rst = orc.find_fundamental_ransac(x1[:n], x2[:n])
            assert ok == rok, (n, seed, ok, rok)
            # 8..13 pairs: the median (element n / 2 < 7 of the sorted errors) is one of the ~1e-20 residuals of the seven SAMPLE points themselves, so which subset wins is
            # decided by rounding noise of the 7-point solver (libm) — in OpenCV too.  With the oracle's libm (emulator) the choice is identical; on the device only what
            # is well defined is compared: success, 300 iterations, and that the returned matrix is an exact 7-point model of the data (>= 7 pairs at ~zero error).
            exact = exact_lmeds or n < 8 or n >= 14
            if n >= 8 and exact: assert (st == rst).all(), (n, seed, st, rst)           # iterations (300 at confidence 0.99), winning iteration / root, inliers
            if ok:
                if exact: assert np.abs(F - rF).max() <= 1e-9 * np.abs(rF).max()
                if n >= 8:
                    lmeds_ok += 1; assert st[0] == rst[0] == 300 and st[3] >= 7 and rst[3] >= 7
                    a = np.c_[x1[:n], np.ones(n)].astype('f8'); b = np.c_[x2[:n], np.ones(n)].astype('f8')
                    l2 = (F @ a.T).T; l1 = (F.T @ b.T).T; e = (b * l2).sum(1) ** 2
                    err = np.maximum(e / (l2[:, 0] ** 2 + l2[:, 1] ** 2), e / (l1[:, 0] ** 2 + l1[:, 1] ** 2))      # FMEstimatorCallback::computeError
                    assert (err <= max(1e-6, 1e-3 * np.median(err) if n >= 14 else 1e-6)).sum() >= 7 or n >= 14
            else:
                assert (F == 0).all()
    assert lmeds_ok >= 8


def check_ransac_batch(lib, orc, xp):
    """batched device form incl. the Frame.cc:454-472 selection rule against the previous frame's boxes"""
    B, cap, mb = 4, 640, 4
    keys = np.zeros((B, cap), KP_DTYPE); n = np.zeros(B, 'i4'); prev = np.zeros((B, cap, 2), 'f4')
    have = np.array([1, 0, 1, 1], 'i4'); nb = np.array([2, 1, 1, 1], 'i4'); boxes = np.zeros((B, mb, 4), 'f4')
    boxes[0, 0] = [100, 80, 200, 220]; boxes[0, 1] = [400, 200, 120, 200]; boxes[1, 0] = [0, 0, 640, 480]; boxes[2, 0] = [-200, -200, 1100, 900]; boxes[3, 0] = [300, 100, 100, 100]
    refs = []
    for b in range(B):
        x1, x2 = two_view(300 + 60 * b, 20 + b, 5)
        if b == 0:                                            # a moving "person": pairs whose previous position is inside box 0 are shifted
            inb = (x2[:, 0] > 100) & (x2[:, 0] < 300) & (x2[:, 1] > 80) & (x2[:, 1] < 300)
            x2[inb] += np.float32(7.0)
        n[b] = len(x1); keys[b, :n[b]]['x'] = x1[:, 0]; keys[b, :n[b]]['y'] = x1[:, 1]; prev[b, :n[b]] = x2
        c, p = orc.fm_select(x1, x2, have[b], boxes[b, :nb[b]])
        if b == 2:
            assert len(c) == n[b]                              # every pair inside the box: at most 20 remain -> all pairs are used (:469-472)
        refs.append(orc.find_fundamental_ransac(c, p))
    d = [xp(a) for a in (keys.view(np.uint8).reshape(B, cap, 28), n, prev, have, boxes, nb)]
    dF = xp(np.zeros((B, 9), 'f8')); dok = xp(np.zeros(B, 'i4')); dst = xp(np.zeros((B, 4), 'i4'))
    fundamental_ransac_batch_dev(lib, B, cap, d[0], d[1], d[2], dF, dok, dst, pre_have=d[3], pre_boxes=d[4], pre_nboxes=d[5], max_boxes=mb)
    g = lambda a: np.asarray(a.cpu() if hasattr(a, 'cpu') else a)
    F, ok, st = g(dF), g(dok), g(dst)
    for b in range(B):
        rok, rF, _, rst = refs[b]
        assert ok[b] == rok == 1 and (st[b] == rst).all(), (b, st[b], rst)
        assert np.abs(F[b].reshape(3, 3) - rF).max() <= 1e-9 * np.abs(rF).max()
