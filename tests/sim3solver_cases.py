"""Shared cases for Sim3Solver (LoopClosing::ComputeSim3's RANSAC initialiser): device (or emulator) against the oracle."""
import ctypes
import numpy as np
from sg_slam_amd.sim3solver import Sim3Solver

K = np.array([535.4, 539.2, 320.1, 247.6], 'f4')


def make_pairs(seed, n=120, outliers=0.3, scale=1.0, noise=0.004):
    """n 3-D correspondences between two keyframes' camera frames related by a similarity X1 = s R X2 + t, with gross outliers"""
    rng = np.random.RandomState(seed)
    w = rng.normal(0, 0.25, 3); th = np.linalg.norm(w); k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(0, 0.3, 3)
    X2 = np.c_[rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(2.0, 6.0, n)]
    X1 = scale * (R @ X2.T).T + t + rng.normal(0, noise, (n, 3))
    bad = rng.rand(n) < outliers
    X1[bad] += rng.normal(0, 0.8, (int(bad.sum()), 3))
    X1[:, 2] = np.maximum(X1[:, 2], 0.5)
    sig2 = (1.2 ** rng.randint(0, 4, n)) ** 2
    e = (9.210 * sig2).astype('f4')
    return X1.astype('f4'), X2.astype('f4'), e, e.copy(), R, t, scale, bad


def check_glibc_rand(orc):
    """the rand() replica against this machine's libc (glibc here): srand(seed); rand() x 2000"""
    libc = ctypes.CDLL(None)
    for seed in (0, 1, 12345):
        libc.srand(seed)
        want = np.array([libc.rand() for _ in range(2000)], 'i8')
        assert (orc.glibc_rand_sequence(seed, 2000).astype('i8') == want).all(), seed


def check_horn(orc):
    """Horn's closed form recovers an exact similarity from three points (oracle self-check, both scale modes)"""
    for seed in range(5):
        X1, X2, _, _, R, t, s, _ = make_pairs(seed, 3, 0.0, 1.0 + 0.1 * seed, 0.0)
        Rr, tr, sr = orc.sim3_horn(X1.T, X2.T, False)
        assert np.abs(Rr - R).max() < 2e-4 and np.abs(tr - t).max() < 2e-3 and abs(sr - s) < 1e-3, (seed, np.abs(Rr - R).max(), np.abs(tr - t).max(), sr, s)


def check_solver(lib, orc, n_cases=4, exact=True):
    found_n = 0
    for c in range(n_cases):
        for fix in (True, False):
            X1, X2, e1, e2, R, t, s, bad = make_pairs(10 + c, 100 + 30 * c, 0.25 + 0.1 * (c % 3), 1.0 if fix else 1.0 + 0.05 * c)
            draws = orc.glibc_rand_sequence(c, 3 * 400)
            S = Sim3Solver(X1, X2, e1, e2, K, K, fix, lib=lib); O = orc.Sim3SolverOracle(X1, X2, e1, e2, K, K, fix)
            S.SetRansacParameters(0.99, 20, 300); O.set_ransac_parameters(0.99, 20, 300)
            assert S.max_iterations() == O.max_iterations()
            used = 0; done = False
            for call in range(80):                                   # LoopClosing calls iterate(5, ...) until a model is found or bNoMore
                gT, gnm, ginl, gni, grun = S.iterate(5, draws[3 * used:3 * used + 15])
                eT, enm, einl, eni, erun = O.iterate(5, draws[3 * used:3 * used + 15])
                assert (gT is None) == (eT is None) and gnm == enm and grun == erun, (c, fix, call, gT is None, eT is None, gnm, enm, grun, erun)
                used += erun
                if eT is not None:
                    if exact: assert (gT == eT).all() and (ginl == einl).all() and gni == eni
                    else: assert np.abs(gT - eT).max() <= 2e-5 * max(1.0, np.abs(eT).max()) and (ginl != einl).sum() <= 1 and abs(gni - eni) <= 1
                    assert eni > 20 and einl.sum() == eni
                    # the model is near the generating similarity (iterate returns at the FIRST model with more than minInliers inliers, so it can be a mediocre one) and its
                    # inliers are true correspondences
                    sR = eT[:3, :3]; assert np.abs(sR - s * R).max() < 0.25 and np.abs(eT[:3, 3] - t).max() < 0.6
                    assert einl[bad].sum() <= 0.1 * eni
                    gR, gt, gs = S.estimate(); eR, et, es = O.estimate()
                    tol = 0 if exact else 2e-5
                    assert np.abs(gR - eR).max() <= tol and np.abs(gt - et).max() <= tol * 10 and abs(gs - es) <= tol
                    found_n += 1; done = True; break
                if enm: done = True; break
            assert done
            S.close(); O.close()
    assert found_n >= n_cases
    # find() = iterate(mRansacMaxIts) with the solver's own rand() replica: equals the oracle fed with glibc's sequence for the same seed
    X1, X2, e1, e2, R, t, s, bad = make_pairs(99, 150, 0.4)
    S = Sim3Solver(X1, X2, e1, e2, K, K, True, rand_seed=7, lib=lib); O = orc.Sim3SolverOracle(X1, X2, e1, e2, K, K, True)
    S.SetRansacParameters(0.99, 20, 300); O.set_ransac_parameters(0.99, 20, 300)
    gT, gnm, ginl, gni, grun = S.find(); eT, enm, einl, eni, erun = O.iterate(O.max_iterations(), orc.glibc_rand_sequence(7, 3 * 300))
    assert (gT is None) == (eT is None) and grun == erun and gnm == enm
    if eT is not None: assert np.abs(gT - eT).max() <= (0 if exact else 2e-5 * np.abs(eT).max())
    # a second call continues the replica's stream where the first stopped (only the iterations that ran consumed draws)
    gT2, _, _, _, grun2 = S.iterate(5); eT2, _, _, _, erun2 = O.iterate(5, orc.glibc_rand_sequence(7, 3 * (erun + 5))[3 * erun:])
    assert (gT2 is None) == (eT2 is None) and grun2 == erun2
    S.close(); O.close()
    # degenerate: fewer correspondences than minInliers -> bNoMore at once; all outliers -> no model, bNoMore after maxIts
    X1, X2, e1, e2, *_ = make_pairs(5, 10, 0.0)
    S = Sim3Solver(X1, X2, e1, e2, K, K, True, lib=lib); S.SetRansacParameters(0.99, 20, 300)
    T, nm, inl, ni, run = S.iterate(5); assert T is None and nm and run == 0 and not inl.any()
    S.close()
    X1, X2, e1, e2, *_ = make_pairs(6, 60, 1.0)
    S = Sim3Solver(X1, X2, e1, e2, K, K, True, lib=lib); O = orc.Sim3SolverOracle(X1, X2, e1, e2, K, K, True)
    S.SetRansacParameters(0.99, 20, 300); O.set_ransac_parameters(0.99, 20, 300)
    d = orc.glibc_rand_sequence(3, 3 * 300)
    gT, gnm, _, _, grun = S.iterate(300, d); eT, enm, _, _, erun = O.iterate(300, d)
    assert gT is None and eT is None and gnm and enm and grun == erun == O.max_iterations()
    S.close(); O.close()
