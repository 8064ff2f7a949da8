"""GPU parity (MI355X) for LocalBundleAdjustment vs the oracle: identical iteration counts and erase flags,
poses / points / final chi2 within 1e-5 relative."""
import numpy as np
import pytest
from scenes import make_ba_problem, CAM
from sg_slam_amd.optimizer import Optimizer
from test_localba import close, points_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed,n_free,n_fixed,n_points', [(11, 8, 5, 400), (12, 3, 0, 120), (13, 16, 10, 900), (14, 20, 40, 2000), (15, 30, 8, 1500), (16, 70, 30, 5000)])
def test_gpu_matches_oracle(gpulib, oracle, seed, n_free, n_fixed, n_points):
    prob, _, _ = make_ba_problem(oracle, n_free=n_free, n_fixed=n_fixed, n_points=n_points, seed=seed)
    eposes, epoints, eerase, etrace, eiters = oracle.local_ba(prob, CAM)
    p2 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in prob.items()}
    erase, stats = Optimizer.LocalBundleAdjustment(p2, CAM, lib=gpulib)
    assert stats['iterations'] == tuple(eiters)
    assert (erase == eerase).all()
    assert close(p2['poses'], eposes) and points_close(p2['points'], epoints)
    ref = etrace[1, eiters[1] - 1, 0]
    assert abs(stats['chi2'][1] - ref) <= 1e-5 * max(1.0, ref)


def test_gpu_stop_flag(gpulib, oracle):
    prob, _, _ = make_ba_problem(oracle, seed=5)
    p2 = {k: (v.copy() if hasattr(v, 'copy') else v) for k, v in prob.items()}
    erase, stats = Optimizer.LocalBundleAdjustment(p2, CAM, stop_flag=np.array([1], 'i4'), lib=gpulib)
    assert (p2['poses'] == prob['poses']).all() and (p2['points'] == prob['points']).all() and erase.sum() == 0
