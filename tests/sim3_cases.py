"""Shared assertions for Optimizer::OptimizeSim3 (tier N4): emulator (CPU tier) and device (-m gpu) against the oracle, and the oracle against the generating similarity."""
import numpy as np
from scenes import CAM

K = np.array([CAM['fx'], CAM['fy'], CAM['cx'], CAM['cy']], 'f4')


def quat_from_rotvec(w):
    th = np.linalg.norm(w)
    if th < 1e-12: return np.array([0, 0, 0, 1.0])
    a = w / th
    return np.r_[a * np.sin(th / 2), np.cos(th / 2)]


def quat_rot(q, v):
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return v @ R.T


def make_problem(seed, n=120, outliers=0.1, noise=0.5, scale=1.0, perturb=0.02):
    """two keyframes seeing the same points; x1 = S12 * x2 with S12 = (q, t, s); observations with pixel noise and gross outliers; a perturbed initial estimate"""
    rng = np.random.RandomState(seed)
    q = quat_from_rotvec(rng.normal(0, 0.15, 3)); t = rng.normal(0, 0.3, 3); s = scale
    P2 = np.c_[rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(2.0, 6.0, n)]
    P1 = s * quat_rot(q, P2) + t
    keep = P1[:, 2] > 0.5
    P1, P2 = P1[keep], P2[keep]; n = len(P1)
    proj = lambda P: np.c_[P[:, 0] / P[:, 2] * K[0] + K[2], P[:, 1] / P[:, 2] * K[1] + K[3]]
    o1 = proj(P1) + rng.normal(0, noise, (n, 2)); o2 = proj(P2) + rng.normal(0, noise, (n, 2))
    bad = rng.rand(n) < outliers
    o1[bad] += rng.uniform(-60, 60, (bad.sum(), 2))
    lv = rng.randint(0, 8, (2, n)); info = (1.0 / 1.2 ** (2 * lv)).astype('f4')
    q0 = quat_from_rotvec(rng.normal(0, perturb, 3)); qi = np.r_[q0[3] * q[:3] + q[3] * q0[:3] + np.cross(q0[:3], q[:3]), q0[3] * q[3] - q0[:3] @ q[:3]]
    S0 = np.r_[qi, t + rng.normal(0, perturb, 3), s * (1 + rng.normal(0, perturb))]
    return dict(p1c=P1.astype('f4'), p2c=P2.astype('f4'), obs1=o1.astype('f4'), obs2=o2.astype('f4'), info1=info[0], info2=info[1], S0=S0, truth=np.r_[q, t, s], bad=bad)


def sim3_close(a, b, tol=1e-5):
    a = np.asarray(a, 'f8'); b = np.asarray(b, 'f8')
    qa, qb = a[:4] / np.linalg.norm(a[:4]), b[:4] / np.linalg.norm(b[:4])
    if qa @ qb < 0: qb = -qb
    return np.abs(qa - qb).max() < tol and np.abs(a[4:7] - b[4:7]).max() < tol * max(1.0, np.abs(b[4:7]).max()) and abs(a[7] - b[7]) < tol * abs(b[7])


def check_oracle_recovers(orc):
    """known answer: from a perturbed start the oracle returns the generating similarity (to the noise level), flags the planted outliers, keeps the scale when told to"""
    for seed in range(4):
        pr = make_problem(seed, outliers=0.1, noise=0.3, scale=[1.0, 1.3, 0.8, 1.0][seed])
        nin, S, inl, it = orc.optimize_sim3(pr['p1c'], pr['p2c'], pr['obs1'], pr['obs2'], pr['info1'], pr['info2'], K, K, pr['S0'], 10.0, False)
        assert nin >= 0.8 * (~pr['bad']).sum() and sim3_close(S, pr['truth'], 2e-2) and it[0] >= 1 and it[1] >= 1
        assert inl[pr['bad']].mean() < 0.1                               # gross outliers are dropped
    pr = make_problem(9, scale=1.0)
    nin, S, inl, it = orc.optimize_sim3(pr['p1c'], pr['p2c'], pr['obs1'], pr['obs2'], pr['info1'], pr['info2'], K, K, pr['S0'], 10.0, True)
    assert S[7] == pr['S0'][7]                                            # bFixScale: update[6] is zeroed in oplus, the scale never moves
    few = {k: (v[:8] if hasattr(v, 'shape') and v.ndim and len(v) > 8 else v) for k, v in pr.items()}
    nin, S, inl, it = orc.optimize_sim3(few['p1c'], few['p2c'], few['obs1'], few['obs2'], few['info1'], few['info2'], K, K, pr['S0'], 10.0, False)
    assert nin == 0 and (S == pr['S0']).all() and it[1] == 0             # fewer than 10 survivors: return 0 before the estimate is read back


def check_sim3(lib, orc, n_cases=6):
    from sg_slam_amd.optimizer import Optimizer
    for seed in range(n_cases):
        pr = make_problem(100 + seed, n=[60, 120, 400, 1000, 30, 200][seed % 6], outliers=[0.0, 0.1, 0.3, 0.05, 0.2, 0.5][seed % 6], scale=[1.0, 1.2, 0.7, 1.0, 1.5, 1.0][seed % 6],
                          perturb=[0.02, 0.05, 0.01, 0.03, 0.02, 0.08][seed % 6])
        for fix in (False, True):
            en, eS, einl, eit = orc.optimize_sim3(pr['p1c'], pr['p2c'], pr['obs1'], pr['obs2'], pr['info1'], pr['info2'], K, K, pr['S0'], 10.0, fix)
            gn, gS, ginl, git = Optimizer.OptimizeSim3(pr['p1c'], pr['p2c'], pr['obs1'], pr['obs2'], pr['info1'], pr['info2'], K, K, pr['S0'], 10.0, fix, lib=lib)
            # the numeric Jacobians (central differences with delta 1e-9) carry ~1e-7 relative noise, so at convergence the accept / stop decisions of the last LM
            # iteration depend on the summation order: iteration counts may differ by one; the estimate and the inlier set do not
            assert np.abs(git - eit).max() <= 1, (seed, fix, git, eit)
            assert gn == en and (ginl == einl).all(), (seed, fix, gn, en, int((ginl != einl).sum()))
            assert sim3_close(gS, eS, 1e-5), (seed, fix, gS, eS)
    # degenerate inputs
    pr = make_problem(7)
    assert Optimizer.OptimizeSim3(pr['p1c'][:0], pr['p2c'][:0], pr['obs1'][:0], pr['obs2'][:0], pr['info1'][:0], pr['info2'][:0], K, K, pr['S0'], 10.0, False, lib=lib)[0] == 0
    gn, gS, ginl, git = Optimizer.OptimizeSim3(pr['p1c'][:8], pr['p2c'][:8], pr['obs1'][:8], pr['obs2'][:8], pr['info1'][:8], pr['info2'][:8], K, K, pr['S0'], 10.0, False, lib=lib)
    assert gn == 0 and (gS == pr['S0']).all()


# ---- essential graph (pose graph over keyframes, EdgeSim3) -----------------------------------------------------------------------
def qmul(a, b):
    return np.r_[a[3] * b[:3] + b[3] * a[:3] + np.cross(a[:3], b[:3]), a[3] * b[3] - a[:3] @ b[:3]]


def s_mul(A, B):       # Sim3 product on (q, t, s) rows
    return np.r_[qmul(A[:4], B[:4]), A[7] * quat_rot(A[:4], B[4:7]) + A[4:7], A[7] * B[7]]


def s_inv(A):
    qc = np.r_[-A[:3], A[3]]
    return np.r_[qc, quat_rot(qc, (-1.0 / A[7]) * A[4:7]), 1.0 / A[7]]


def make_graph(seed, nv=40, noise=0.02, scale_drift=0.0):
    """keyframes on a loop; exact relative measurements (spanning chain + covisibility shortcuts + one loop edge); initial estimates = the truth with accumulated drift, vertex 0 fixed"""
    rng = np.random.RandomState(seed)
    truth = []
    for i in range(nv):
        a = 2 * np.pi * i / nv
        q = quat_from_rotvec(np.array([0.0, a, 0.0]) + rng.normal(0, 0.02, 3)); t = np.array([2 * np.cos(a), 0.1 * rng.randn(), 2 * np.sin(a)])
        truth.append(np.r_[q, t, 1.0])
    truth = np.array(truth)
    ei, ej = [], []
    for i in range(1, nv): ei.append(i); ej.append(i - 1)                        # spanning tree: child -> parent
    for i in range(3, nv, 2): ei.append(i); ej.append(i - 3)                      # covisibility
    ei.append(nv - 1); ej.append(0)                                              # the loop
    for i in range(10, nv, 7): ei.append(i); ej.append(i - 10)
    ei, ej = np.array(ei, 'i4'), np.array(ej, 'i4')
    meas = np.array([s_mul(truth[j], s_inv(truth[i])) for i, j in zip(ei, ej)])    # Sji = Sjw * Swi
    S0 = truth.copy()
    drift = np.zeros(6)
    for i in range(1, nv):                                                        # accumulated drift along the chain
        drift = drift + rng.normal(0, noise, 6)
        d = np.r_[quat_from_rotvec(drift[:3] * 0.3), drift[3:], np.exp(scale_drift * i / nv)]
        S0[i] = s_mul(d, truth[i])
    fixed = np.zeros(nv, np.uint8); fixed[0] = 1
    return dict(S0=S0, truth=truth, fixed=fixed, ei=ei, ej=ej, meas=meas)


def check_eg_oracle_recovers(orc):
    """known answer: exact relative measurements -> chi2 collapses and every keyframe returns to the generating pose (bFixScale = true, the RGB-D setting, Tracking / LoopClosing
    pass mSensor != MONOCULAR).  With a free scale the vendored g2o stalls once the rotation part of a step falls below 1e-5 while its scale part is above it: that branch of
    Sim3's exponential map carries B = ((sigma^2/2 - sigma + 1) s) / sigma^3 (sim3.h:110), which diverges like 1/sigma^3 instead of tending to 1/6 — restated as written, so the
    oracle stalls the same way; the test then only asks for a decrease."""
    for seed, (fix, sd) in enumerate(((True, 0.0), (True, 0.0), (False, 0.0), (False, 0.15))):
        g = make_graph(seed, 30, scale_drift=sd)
        S, st = orc.optimize_essential_graph(g['S0'], g['fixed'], g['ei'], g['ej'], g['meas'], fix, 20)
        assert (S[0] == g['S0'][0]).all()                                          # the fixed vertex is not touched
        if fix:
            assert st[1] > 1e-3 and st[2] < 1e-9 * st[1] and st[0] >= 3
            for v in range(len(S)):
                assert sim3_close(S[v], g['truth'][v], 1e-6), (seed, v)
        else:
            assert st[2] < 0.05 * st[1]
    pts = np.random.RandomState(3).uniform(-3, 3, (50, 3)).astype('f4'); ref = np.random.RandomState(4).randint(0, 30, 50).astype('i4')
    inv = np.array([s_inv(s) for s in S])
    out = orc.correct_map_points(pts, ref, g['S0'], inv)
    exp = np.array([inv[r][7] * quat_rot(inv[r][:4], g['S0'][r][7] * quat_rot(g['S0'][r][:4], p) + g['S0'][r][4:7]) + inv[r][4:7] for p, r in zip(pts.astype('f8'), ref)])
    assert np.abs(out - exp).max() < 1e-5


def check_eg(lib, orc, n_cases=4):
    from sg_slam_amd.optimizer import Optimizer
    for seed in range(n_cases):
        g = make_graph(50 + seed, [20, 60, 150, 400][seed % 4], noise=[0.02, 0.01, 0.03, 0.005][seed % 4], scale_drift=[0.0, 0.1, 0.0, 0.05][seed % 4])
        for fix in (True, False):
            eS, est = orc.optimize_essential_graph(g['S0'], g['fixed'], g['ei'], g['ej'], g['meas'], fix, 20)
            gS, gst = Optimizer.OptimizeEssentialGraph(g['S0'], g['fixed'], g['ei'], g['ej'], g['meas'], fix, 20, lib=lib)
            # once chi2 has collapsed to rounding level (exact synthetic measurements) further accept / stop decisions are noise: compare the iteration counts only before that
            if est[2] > 1e-12 * est[1]: assert abs(gst[0] - est[0]) <= 1, (seed, fix, gst, est)
            else: assert gst[2] <= 1e-9 * gst[1], (seed, fix, gst, est)
            assert abs(gst[1] - est[1]) <= 1e-9 * est[1]
            # bFixScale = false (monocular only; SG-SLAM's RGB-D setting passes true): the optimisation stalls in the vendored Sim3 exponential's diverging branch (see
            # check_eg_oracle_recovers) and where exactly it stalls depends on rounding — only a loose agreement is defined there
            for v in range(len(eS)):
                assert sim3_close(gS[v], eS[v], 1e-5 if fix else 2e-3), (seed, fix, v, gS[v], eS[v])
    pts = np.random.RandomState(5).uniform(-3, 3, (300, 3)).astype('f4'); ref = np.random.RandomState(6).randint(0, len(eS), 300).astype('i4')
    inv = np.array([s_inv(s) for s in eS])
    assert np.abs(Optimizer.CorrectMapPoints(pts, ref, g['S0'], inv, lib=lib) - orc.correct_map_points(pts, ref, g['S0'], inv)).max() <= 1e-6
