"""Known-answer tests that pin the ORACLE's solver pieces against independent arithmetic (the reference ships no golden vectors for
this path): analytic Jacobians of the four g2o edge types against central differences of the oracle's own error function, and the Schur-complement
solve against a dense numpy solve of the full damped normal equations assembled from the same blocks."""
import ctypes as C
import numpy as np
import pytest
from scenes import make_ba_problem, CAM

P = lambda a: a.ctypes.data_as(C.c_void_p)
D = C.c_double


def _cam():           # the intrinsics as the float32 values the cv::Mat boundary carries
    return [D(float(np.float32(CAM[k]))) for k in ('fx', 'fy', 'cx', 'cy', 'bf')]


def _pose(rng):
    w = rng.randn(3) * 0.3; th = np.linalg.norm(w); K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    T = np.eye(4, dtype='f4'); T[:3, :3] = R; T[:3, 3] = rng.randn(3) * 0.5
    return T


@pytest.mark.parametrize('stereo', [0, 1])
def test_pose_edge_jacobian_vs_central_differences(oracle, stereo):
    """Edge(Stereo)SE3ProjectXYZOnlyPose: d err / d delta at delta = 0 for T <- exp(delta) T (types_six_dof_expmap.cpp:266-288, 335-364).  The stereo projection
    rounds 1/z to float (.cpp:299-306), which puts ~1e-5 px steps into the error: its difference quotient uses a larger step and a looser bound."""
    L = oracle.lib(); rng = np.random.RandomState(3 + stereo)
    h, tol = (1e-3, 2e-3) if stereo else (1e-6, 1e-6)
    for _ in range(50):
        T = _pose(rng); Xc = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1, 1), rng.uniform(1.0, 6.0)])
        Xw = (T[:3, :3].astype('f8').T @ (Xc - T[:3, 3].astype('f8')))
        obs = rng.randn(3) * 3 + np.array([320, 240, 300.0])
        err = np.zeros(3); J = np.zeros(18); z6 = np.zeros(6)
        L.orc_kat_pose_edge(P(T), P(Xw), P(obs), stereo, *_cam(), P(z6), P(err), P(J))
        J = J.reshape(3, 6)
        for a in range(6):
            ep = np.zeros(3); em = np.zeros(3); dj = np.zeros(18)
            d = np.zeros(6); d[a] = h; L.orc_kat_pose_edge(P(T), P(Xw), P(obs), stereo, *_cam(), P(d), P(ep), P(dj))
            d[a] = -h; L.orc_kat_pose_edge(P(T), P(Xw), P(obs), stereo, *_cam(), P(d), P(em), P(dj))
            num = (ep - em) / (2 * h)
            rows = 3 if stereo else 2
            assert np.abs(num[:rows] - J[:rows, a]).max() <= tol * max(1.0, np.abs(J[:rows]).max()), (a, num, J[:, a])


@pytest.mark.parametrize('stereo', [0, 1])
def test_ba_edge_jacobians_vs_central_differences(oracle, stereo):
    """Edge(Stereo)SE3ProjectXYZ: d err / d pose increment (3x6) and d err / d point (3x3), types_six_dof_expmap.cpp:103-139, 188-234"""
    L = oracle.lib(); rng = np.random.RandomState(13 + stereo)
    h, tol = (1e-3, 2e-3) if stereo else (1e-6, 1e-6)
    for _ in range(50):
        T = _pose(rng); Xc = np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1, 1), rng.uniform(1.0, 6.0)])
        Xw = (T[:3, :3].astype('f8').T @ (Xc - T[:3, 3].astype('f8')))
        obs = rng.randn(3) * 3 + np.array([320, 240, 300.0])
        def f(dp, dx):
            err = np.zeros(3); Jp = np.zeros(18); Jx = np.zeros(9)
            L.orc_kat_ba_edge(P(T), P(Xw), P(obs), stereo, *_cam(), P(np.ascontiguousarray(dp, 'f8')), P(np.ascontiguousarray(dx, 'f8')), P(err), P(Jp), P(Jx))
            return err, Jp.reshape(3, 6), Jx.reshape(3, 3)
        _, Jp, Jx = f(np.zeros(6), np.zeros(3))
        rows = 3 if stereo else 2
        scale = max(1.0, np.abs(Jp[:rows]).max(), np.abs(Jx[:rows]).max())
        for a in range(6):
            d = np.zeros(6); d[a] = h
            num = (f(d, np.zeros(3))[0] - f(-d, np.zeros(3))[0]) / (2 * h)
            assert np.abs(num[:rows] - Jp[:rows, a]).max() <= tol * scale
        for a in range(3):
            d = np.zeros(3); d[a] = h
            num = (f(np.zeros(6), d)[0] - f(np.zeros(6), -d)[0]) / (2 * h)
            assert np.abs(num[:rows] - Jx[:rows, a]).max() <= tol * scale


@pytest.mark.parametrize('robust', [0, 1])
def test_schur_step_solves_the_full_normal_equations(oracle, robust):
    """BlockSolver::solve (block_solver.hpp:367-486): the (xp, xl) the oracle gets through the Schur complement + landmark back-substitution must solve the
    full system [[Hpp + l I, Hpl], [Hpl^T, Hll + l I]] [xp; xl] = [bp; bl] assembled densely from the same blocks — and the blocks must be the sums of
    J^T W J over the edges, recomputed here from the per-edge Jacobians of the tap above."""
    L = oracle.lib()
    prob, _, _ = make_ba_problem(oracle, n_free=6, n_fixed=3, n_points=120, seed=21)
    poses = np.ascontiguousarray(prob['poses'], 'f4').reshape(-1, 16); fixed = np.ascontiguousarray(prob['pose_fixed'], np.uint8)
    pts = np.ascontiguousarray(prob['points'], 'f4'); ep = np.ascontiguousarray(prob['edge_pose'], 'i4'); el = np.ascontiguousarray(prob['edge_point'], 'i4')
    eo = np.ascontiguousarray(prob['edge_obs'], 'f4'); ei = np.ascontiguousarray(prob['edge_info'], 'f4')
    npz, nl, ne = len(poses), len(pts), len(ep)
    nf = int((fixed == 0).sum()); NP = 6 * nf; lam = 3.7
    Hpp = np.zeros(nf * 36); bp = np.zeros(NP); Hll = np.zeros(nl * 9); bl = np.zeros(nl * 3); Hpl = np.zeros(ne * 18); xp = np.zeros(NP); xl = np.zeros(nl * 3)
    hidx = np.zeros(npz, 'i4')
    L.orc_kat_ba_step.restype = C.c_int
    ok = L.orc_kat_ba_step(npz, P(poses), P(fixed), nl, P(pts), ne, P(ep), P(el), P(eo), P(ei), C.c_float(CAM['fx']), C.c_float(CAM['fy']), C.c_float(CAM['cx']),
                           C.c_float(CAM['cy']), C.c_float(CAM['bf']), D(lam), robust, P(Hpp), P(bp), P(Hll), P(bl), P(Hpl), P(xp), P(xl), P(hidx))
    assert ok == 1
    N = NP + 3 * nl
    H = np.zeros((N, N)); b = np.concatenate([bp, bl])
    for i in range(nf): H[6 * i:6 * i + 6, 6 * i:6 * i + 6] = Hpp[36 * i:36 * i + 36].reshape(6, 6)
    for l in range(nl): H[NP + 3 * l:NP + 3 * l + 3, NP + 3 * l:NP + 3 * l + 3] = Hll[9 * l:9 * l + 9].reshape(3, 3)
    for k in range(ne):
        i = hidx[ep[k]]
        if i < 0: continue
        blk = Hpl[18 * k:18 * k + 18].reshape(6, 3)
        H[6 * i:6 * i + 6, NP + 3 * el[k]:NP + 3 * el[k] + 3] += blk; H[NP + 3 * el[k]:NP + 3 * el[k] + 3, 6 * i:6 * i + 6] += blk.T
    Hd = H + lam * np.eye(N)
    seen = np.zeros(nl, bool); seen[el] = True          # landmarks without an edge are not vertices of the optimisation
    act = np.concatenate([np.ones(NP, bool), np.repeat(seen, 3)])
    x = np.concatenate([xp, xl])
    xd = np.linalg.solve(Hd[np.ix_(act, act)], b[act])
    assert np.abs(x[act] - xd).max() <= 1e-9 * max(1.0, np.abs(xd).max())
    # the blocks themselves: H = sum J^T (rho1 info) J, b = -sum J^T (rho1 info) err, from the per-edge tap (Huber weight recomputed here)
    H2 = np.zeros((N, N)); b2 = np.zeros(N)
    for k in range(ne):
        T = poses[ep[k]]; X = pts[el[k]].astype('f8'); obs = eo[k].astype('f8'); stereo = int(not (eo[k, 2] < 0))
        err = np.zeros(3); Jp = np.zeros(18); Jx = np.zeros(9)
        L.orc_kat_ba_edge(P(np.ascontiguousarray(T)), P(X), P(obs), stereo, *_cam(), P(np.zeros(6)), P(np.zeros(3)), P(err), P(Jp), P(Jx))
        rows = 3 if stereo else 2
        Jp = Jp.reshape(3, 6)[:rows]; Jx = Jx.reshape(3, 3)[:rows]; e = err[:rows]; info = float(ei[k])
        rho1 = 1.0
        if robust:
            c2 = info * float(e @ e); delta = float(np.float32(np.sqrt(7.815 if stereo else 5.991)))
            if c2 > delta * delta: rho1 = delta / np.sqrt(c2)
        w = rho1 * info
        i = hidx[ep[k]]; s = NP + 3 * el[k]
        H2[s:s + 3, s:s + 3] += w * Jx.T @ Jx; b2[s:s + 3] -= w * Jx.T @ e
        if i >= 0:
            H2[6 * i:6 * i + 6, 6 * i:6 * i + 6] += w * Jp.T @ Jp; b2[6 * i:6 * i + 6] -= w * Jp.T @ e
            H2[6 * i:6 * i + 6, s:s + 3] += w * Jp.T @ Jx; H2[s:s + 3, 6 * i:6 * i + 6] += w * Jx.T @ Jp
    assert np.abs(H2 - H).max() <= 1e-9 * np.abs(H).max() and np.abs(b2 - b).max() <= 1e-9 * max(1.0, np.abs(b).max())
