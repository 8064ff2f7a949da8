"""PoseOptimization: oracle known-answer tests (the reference ships none) + emulator parity."""
import numpy as np
import pytest
from scenes import make_pose_problem, CAM
from sg_slam_amd.optimizer import Optimizer

REL = 1e-5     # north star: pose / residuals within 1e-5 relative


def pose_close(Ta, Tb, rel=REL):
    return np.abs(Ta - Tb).max() <= rel * max(1.0, np.abs(Tb).max())


def test_oracle_recovers_generating_pose(oracle):
    # noise-free, no outliers: LM must return the generating pose and chi2 -> 0 (SURVEY §8(c) KAT)
    frame, Ttrue, _ = make_pose_problem(oracle, n=300, seed=1, outlier_frac=0.0, noise_px=0.0, init_sigma=0.03)
    is2 = oracle.orb_params()['inv_sigma2']
    n, T, out, trace, tn = oracle.pose_optimization(frame, CAM, is2, want_trace=True)
    assert n == int(frame['has_mp'].sum()) and out.sum() == 0
    assert np.abs(T - Ttrue).max() < 2e-4          # float32 observations / points limit the exactness
    assert trace[3, tn[3] - 1, 0] < 1e-2 * n


def test_oracle_rejects_gross_outliers(oracle):
    frame, Ttrue, gross = make_pose_problem(oracle, n=600, seed=2)
    is2 = oracle.orb_params()['inv_sigma2']
    n, T, out = oracle.pose_optimization(frame, CAM, is2)
    has = frame['has_mp'] > 0
    assert out[has & gross].mean() > 0.9           # gross outliers flagged
    assert out[has & ~gross].mean() < 0.15
    assert np.abs(T - Ttrue).max() < 0.02


def test_oracle_few_correspondences(oracle):
    frame, _, _ = make_pose_problem(oracle, n=2, seed=3)
    frame['has_mp'][:] = 1
    n, T, out = oracle.pose_optimization(frame, CAM, oracle.orb_params()['inv_sigma2'])
    assert n == 0 and (T == frame['Tcw']).all()    # Optimizer.cc:364-365


@pytest.mark.parametrize('n,seed', [(200, 42), (400, 43), (800, 44), (60, 45), (9, 46)])
def test_emu_matches_oracle(emu, oracle, n, seed):
    frame, _, _ = make_pose_problem(oracle, n=n, seed=seed)
    is2 = oracle.orb_params()['inv_sigma2']
    en, eT, eout = oracle.pose_optimization(frame, CAM, is2)
    f2 = dict(frame)
    gn = Optimizer.PoseOptimization(f2, CAM, is2, lib=emu)
    assert gn == en
    assert (f2['outlier'] == eout).all()
    assert pose_close(f2['Tcw'], eT)


def test_emu_all_mono_and_all_stereo(emu, oracle):
    is2 = oracle.orb_params()['inv_sigma2']
    for mono_frac in (0.0, 1.0):
        frame, _, _ = make_pose_problem(oracle, n=300, seed=7, mono_frac=mono_frac)
        en, eT, eout = oracle.pose_optimization(frame, CAM, is2)
        f2 = dict(frame)
        gn = Optimizer.PoseOptimization(f2, CAM, is2, lib=emu)
        assert gn == en and (f2['outlier'] == eout).all() and pose_close(f2['Tcw'], eT)


@pytest.mark.parametrize('threads', [64, 256])
@pytest.mark.parametrize('n,seed', [(400, 43), (1000, 47), (9, 46)])
def test_threads_per_frame_variants_emu(emu, oracle, n, seed, threads):
    """one wave per frame (large batches) and four waves per frame (small batches) both reproduce the oracle"""
    frame, _, _ = make_pose_problem(oracle, n=n, seed=seed)
    is2 = oracle.orb_params()['inv_sigma2']
    en, eT, eout = oracle.pose_optimization(frame, CAM, is2)
    emu.tap('sgx_pose_opt_debug_set_threads')(threads)
    try:
        f2 = dict(frame)
        gn = Optimizer.PoseOptimization(f2, CAM, is2, lib=emu)
    finally:
        emu.tap('sgx_pose_opt_debug_set_threads')(0)
    assert gn == en and (f2['outlier'] == eout).all() and pose_close(f2['Tcw'], eT)
