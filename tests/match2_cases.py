"""Shared assertions for the LocalMapping matcher gates (tier N2): Hamming matrix and SearchForTriangulation against the oracle."""
import numpy as np
from scenes import CAM
from sg_slam_amd import synth
from sg_slam_amd.matcher import ORBmatcher


def check_hamming(lib, orc):
    rng = np.random.RandomState(0)
    for na, nb in ((1, 1), (17, 33), (300, 257)):
        a = rng.randint(0, 256, (na, 32)).astype(np.uint8); b = rng.randint(0, 256, (nb, 32)).astype(np.uint8)
        b[: min(na, nb)] = a[: min(na, nb)]                                # zeros on part of the diagonal
        got = ORBmatcher(lib=lib).HammingMatrix(a, b); ref = orc.hamming_matrix(a, b)
        assert (got == ref).all()
        assert ref[0, 0] == 0 and (ref == np.unpackbits(a[:, None, :] ^ b[None, :, :], axis=2).sum(2)).all()     # known answer: naive popcount


def make_keyframes(orc, seed, t1, t2, only_some_mp=True):
    """two keyframes of a layered synthetic stream with a vocabulary-like grouping: node = coarse quantisation of the descriptor's first bits (any function of the
    descriptor works: the algorithm only needs equal keys for features that may match), plus random has_mp / mono flags"""
    gen = synth.LayeredStream(seed=1234 + seed)
    rng = np.random.RandomState(seed)
    out = []
    for t in (t1, t2):
        g, dep, T = gen.frame(t)
        k, d = orc.orb_extract(g)
        ur, z = orc.compute_stereo_from_rgbd(k, dep, CAM['bf'], CAM['depth_factor'])
        ur = ur.copy(); ur[rng.rand(len(k)) < 0.3] = -1                     # some monocular keypoints
        node = ((d[:, 0].astype('i4') & 0x1F) * 7 + (k['octave'] // 3)).astype('i4')      # ~ 90 nodes, descriptor dependent
        node[rng.rand(len(k)) < 0.02] = -1
        has = (rng.rand(len(k)) < (0.6 if only_some_mp else 0.0)).astype(np.uint8)
        Tf = T.astype('f4')
        out.append(dict(keys=k, desc=d, uright=ur, has_mp=has, feat_node=node, Tcw=Tf, cam_center=(-(Tf[:3, :3].T @ Tf[:3, 3])).astype('f4')))
    return gen, out[0], out[1]


def fundamental_12(kf1, kf2):
    """LocalMapping::ComputeF12: F12 = K1^-T t12x R12 K2^-1 (LocalMapping.cc:562-580), float"""
    K = np.array([[CAM['fx'], 0, CAM['cx']], [0, CAM['fy'], CAM['cy']], [0, 0, 1.0]])
    R1, t1 = kf1['Tcw'][:3, :3].astype('f8'), kf1['Tcw'][:3, 3].astype('f8'); R2, t2 = kf2['Tcw'][:3, :3].astype('f8'), kf2['Tcw'][:3, 3].astype('f8')
    R12 = R1 @ R2.T; t12 = -R12 @ t2 + t1
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    return (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype('f4')


def check_triangulation(lib, orc, n_cases=6):
    sf = orc.orb_params()['scale']; sg = orc.orb_params()['sigma2']
    total = 0
    for c in range(n_cases):
        gen, kf1, kf2 = make_keyframes(orc, c, 10 + 3 * c, 14 + 3 * c, only_some_mp=(c % 2 == 0))
        F12 = fundamental_12(kf1, kf2)
        for only_stereo in (False, True):
            for ori in (True, False):
                en, ep = orc.search_for_triangulation(kf1, kf2, F12, only_stereo, CAM, sf, sg, check_ori=ori)
                gn, gp = ORBmatcher(0.6, ori, lib=lib).SearchForTriangulation(kf1, kf2, F12, only_stereo, CAM, sf, sg)
                assert gn == en == len(ep) and (gp == ep).all(), (c, only_stereo, ori, gn, en)
                total += en
                if len(ep):
                    assert (np.diff(ep[:, 0]) > 0).all() and len(set(ep[:, 1])) == len(ep)       # ascending idx1, every idx2 used once
                    assert not kf1['has_mp'][ep[:, 0]].any() and not kf2['has_mp'][ep[:, 1]].any()
                    assert (kf1['feat_node'][ep[:, 0]] == kf2['feat_node'][ep[:, 1]]).all()
    assert total > 200
    # degenerate inputs: an empty keyframe, no common node
    _, kf1, kf2 = make_keyframes(orc, 99, 5, 8)
    e = {k: (v[:0] if hasattr(v, '__len__') and k not in ('Tcw', 'cam_center') else v) for k, v in kf2.items()}
    assert ORBmatcher(lib=lib).SearchForTriangulation(kf1, e, fundamental_12(kf1, kf2), False, CAM, sf, sg)[0] == 0
    kf2b = dict(kf2); kf2b['feat_node'] = kf2['feat_node'] + 100000
    assert ORBmatcher(lib=lib).SearchForTriangulation(kf1, kf2b, fundamental_12(kf1, kf2), False, CAM, sf, sg)[0] == 0
