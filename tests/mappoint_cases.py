"""Shared cases for the batched MapPoint post-steps (UpdateNormalAndDepth, ComputeDistinctiveDescriptors): device (or emulator) against the oracle."""
import numpy as np
from sg_slam_amd import mappoint


def make_points(orc, seed, n=3000, max_obs=40):
    rng = np.random.RandomState(seed)
    sf = orc.orb_params()['scale']
    nobs = rng.randint(0, max_obs + 1, n); nobs[rng.rand(n) < 0.05] = 0; nobs[:3] = (1, 2, max_obs)
    st = np.zeros(n + 1, 'i4'); st[1:] = np.cumsum(nobs)
    xw = np.c_[rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(1, 8, n)].astype('f4')
    oc = rng.normal(0, 0.8, (st[-1], 3)).astype('f4'); rc = rng.normal(0, 0.8, (n, 3)).astype('f4'); rl = rng.randint(0, 8, n).astype('i4')
    base = rng.randint(0, 256, (n, 32)).astype(np.uint8)
    desc = np.repeat(base, nobs, axis=0)
    flip = rng.rand(len(desc), 256) < rng.uniform(0.0, 0.15, (len(desc), 1))          # every observation = the point's descriptor with its own noise level
    desc = np.packbits(np.unpackbits(desc, axis=1) ^ flip.astype(np.uint8), axis=1)
    return dict(sf=sf, st=st, xw=xw, oc=oc, rc=rc, rl=rl, desc=desc, nobs=nobs)


def check_mappoint(lib, orc, n_cases=3):
    for c in range(n_cases):
        P = make_points(orc, 700 + c, 3000 if c else 500, 40 if c < 2 else 150)
        n = len(P['xw'])
        nr0 = np.full((n, 3), 7.0, 'f4'); mn0 = np.full(n, 8.0, 'f4'); mx0 = np.full(n, 9.0, 'f4')
        en, emn, emx = orc.update_normal_and_depth(P['xw'], P['st'], P['oc'], P['rc'], P['rl'], P['sf'], nr0, mn0, mx0)
        gn, gmn, gmx = mappoint.UpdateNormalAndDepth(P['xw'], P['st'], P['oc'], P['rc'], P['rl'], P['sf'], nr0, mn0, mx0, lib=lib)
        assert (gn == en).all() and (gmn == emn).all() and (gmx == emx).all(), c
        none = P['nobs'] == 0
        assert (en[none] == 7.0).all() and (emn[none] == 8.0).all() and (emx[none] == 9.0).all() and none.any()       # untouched
        assert (np.linalg.norm(en[~none], axis=1) <= 1.0 + 1e-5).all() and (emx[~none] >= emn[~none]).all()
        eb = orc.distinctive_descriptors(P['st'], P['desc'])
        gb, gd = mappoint.ComputeDistinctiveDescriptors(P['st'], P['desc'], lib=lib)
        assert (gb == eb).all(), (c, int((gb != eb).sum()))
        assert (eb[none] == -1).all() and (eb[~none] >= 0).all() and (eb[~none] < P['nobs'][~none]).all()
        sel = np.nonzero(~none)[0]
        assert (gd[sel] == P['desc'][P['st'][sel] + eb[sel]]).all()
    # ties: identical descriptors -> the first one; two observations -> index 0 (median of {0, d} at position 0 is 0 for both rows)
    d = np.tile(np.arange(32, dtype=np.uint8), (5, 1))
    assert mappoint.ComputeDistinctiveDescriptors([0, 5], d, lib=lib)[0][0] == 0 == orc.distinctive_descriptors([0, 5], d)[0]
    assert mappoint.ComputeDistinctiveDescriptors(np.zeros(1, 'i4'), np.zeros((0, 32), np.uint8), lib=lib)[0].shape == (0,)


def check_triangulation_step(lib, orc, n_cases=4, exact=True):
    """CreateNewMapPoints' per-pair body on the pairs of SearchForTriangulation between two keyframes of a synthetic stream: every branch (SVD triangulation, stereo
    unprojection of either side, the rejections) against the oracle"""
    import match2_cases as mc
    from scenes import CAM
    from sg_slam_amd.matcher import ORBmatcher
    sf = orc.orb_params()['scale']; sg = orc.orb_params()['sigma2']
    tot = 0; svd_like = 0; branches = set()
    for c in range(n_cases):
        gen, kf1, kf2 = mc.make_keyframes(orc, 800 + c, 6 + c, 12 + 2 * c, only_some_mp=False)
        F12 = mc.fundamental_12(kf1, kf2)
        _, pairs = ORBmatcher(0.6, False, lib=lib).SearchForTriangulation(kf1, kf2, F12, False, CAM, sf, sg)
        rng = np.random.RandomState(c)
        for k in (kf1, kf2):
            ur = k['uright']; z = np.where(ur >= 0, CAM['bf'] / np.maximum(k['keys']['x'] - ur, 1e-3), -1.0).astype('f4')        # mvDepth consistent with mvuRight
            k['depth'] = z; k['keys_un'] = k['keys']
        A = dict(keys_un=kf1['keys'], uright=kf1['uright'], depth=kf1['depth'], Tcw=kf1['Tcw']); B = dict(keys_un=kf2['keys'], uright=kf2['uright'], depth=kf2['depth'], Tcw=kf2['Tcw'])
        for variant in range(3):
            a, b = dict(A), dict(B)
            if variant == 1: a['uright'] = np.full(len(A['uright']), -1, 'f4'); b['uright'] = np.full(len(B['uright']), -1, 'f4')          # monocular pairs only: SVD or nothing
            if variant == 2: b['Tcw'] = B['Tcw'].copy(); b['Tcw'][:3, 3] += rng.normal(0, 0.02, 3).astype('f4')                        # a slightly wrong neighbour pose: more rejections
            en, eok, ex = orc.triangulate_pairs(pairs, a, b, CAM, sf, sg)
            gn, gok, gx = mappoint.TriangulateNewMapPoints(pairs, a, b, CAM, sf, sg, lib=lib)
            if exact:
                assert gn == en and (gok == eok).all() and (gx == ex).all(), (c, variant, gn, en, int((gok != eok).sum()))
            else:
                flips = int((gok != eok).sum()); both = gok & eok
                assert flips <= max(1, len(pairs) // 100), (c, variant, flips, len(pairs))
                assert np.abs(gx[both] - ex[both]).max() <= 2e-3 * max(1.0, np.abs(ex[both]).max()), (c, variant, np.abs(gx[both] - ex[both]).max())
            tot += en; branches.add((variant, en > 0, en < len(pairs)))
            if variant == 1: svd_like += en
    assert tot > 300 and svd_like > 20 and len(branches) >= 3, (tot, svd_like, branches)
    assert mappoint.TriangulateNewMapPoints(np.zeros((0, 2), 'i4'), A, B, CAM, sf, sg, lib=lib)[0] == 0
