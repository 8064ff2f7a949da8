"""GPU parity (MI355X) for PoseOptimization: pose within 1e-5 relative of the oracle, identical
inlier/outlier classification and inlier count."""
import numpy as np
import pytest
from scenes import make_pose_problem, CAM
from sg_slam_amd.optimizer import Optimizer
from test_poseopt import pose_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,seed', [(200, 42), (400, 43), (800, 44), (1000, 47), (60, 45), (9, 46), (2, 48)])
def test_gpu_matches_oracle(gpulib, oracle, n, seed):
    frame, _, _ = make_pose_problem(oracle, n=n, seed=seed)
    is2 = oracle.orb_params()['inv_sigma2']
    en, eT, eout = oracle.pose_optimization(frame, CAM, is2)
    f2 = dict(frame)
    gn = Optimizer.PoseOptimization(f2, CAM, is2, lib=gpulib)
    assert gn == en and (f2['outlier'] == eout).all() and pose_close(f2['Tcw'], eT)


def test_gpu_without_fma_contraction_matches_oracle(gpulib_nofma, oracle):
    """The product fuses multiply-adds in the fp64 solvers; the reference (g2o, plain -O3 on x86-64) does not.  The same sources built without contraction
    must reproduce the oracle (-ffp-contract=off as well) at least as closely: same inlier count and flags, pose within the 1e-5 bar."""
    frame, _, _ = make_pose_problem(oracle, n=800, seed=44)
    is2 = oracle.orb_params()['inv_sigma2']
    en, eT, eout = oracle.pose_optimization(frame, CAM, is2)
    f2 = dict(frame)
    gn = Optimizer.PoseOptimization(f2, CAM, is2, lib=gpulib_nofma)
    assert gn == en and (f2['outlier'] == eout).all() and pose_close(f2['Tcw'], eT)


def test_gpu_mono_stereo(gpulib, oracle):
    is2 = oracle.orb_params()['inv_sigma2']
    for mono_frac in (0.0, 1.0):
        frame, _, _ = make_pose_problem(oracle, n=300, seed=7, mono_frac=mono_frac)
        en, eT, eout = oracle.pose_optimization(frame, CAM, is2)
        f2 = dict(frame)
        gn = Optimizer.PoseOptimization(f2, CAM, is2, lib=gpulib)
        assert gn == en and (f2['outlier'] == eout).all() and pose_close(f2['Tcw'], eT)


@pytest.mark.parametrize('threads', [64, 256])
@pytest.mark.parametrize('n,seed', [(400, 43), (1000, 47), (9, 46)])
def test_threads_per_frame_variants_gpulib(gpulib_taps, oracle, n, seed, threads):
    """one wave per frame (large batches) and four waves per frame (small batches) both reproduce the oracle"""
    frame, _, _ = make_pose_problem(oracle, n=n, seed=seed)
    is2 = oracle.orb_params()['inv_sigma2']
    en, eT, eout = oracle.pose_optimization(frame, CAM, is2)
    gpulib = gpulib_taps
    gpulib.tap('sgx_pose_opt_debug_set_threads')(threads)
    try:
        f2 = dict(frame)
        gn = Optimizer.PoseOptimization(f2, CAM, is2, lib=gpulib)
    finally:
        gpulib.tap('sgx_pose_opt_debug_set_threads')(0)
    assert gn == en and (f2['outlier'] == eout).all() and pose_close(f2['Tcw'], eT)
