"""Optimizer::BundleAdjustment (N4) on the device against the oracle."""
import pytest
import ba_global_cases as bc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n_kf,n_points,seed,n_iter,robust', [(6, 300, 21, 10, True), (12, 700, 22, 20, False), (25, 1200, 23, 5, True), (60, 3000, 24, 10, True), (140, 6000, 25, 10, True)])
def test_gba_gpu(gpulib, oracle, n_kf, n_points, seed, n_iter, robust):
    bc.check_gba(gpulib, oracle, n_kf, n_points, seed, n_iter, robust, outlier_frac=0.05 if robust else 0.0)
