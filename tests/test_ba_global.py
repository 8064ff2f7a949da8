"""Optimizer::BundleAdjustment (N4): oracle known answers + emulator parity."""
import pytest
import ba_global_cases as bc


def test_oracle_robust_flag(oracle):
    bc.check_gba_robust_matters(oracle)


@pytest.mark.parametrize('n_kf,n_points,seed,n_iter,robust', [(6, 300, 21, 10, True), (12, 700, 22, 20, False), (25, 1200, 23, 5, True)])
def test_gba_emu(emu, oracle, n_kf, n_points, seed, n_iter, robust):
    bc.check_gba(emu, oracle, n_kf, n_points, seed, n_iter, robust, outlier_frac=0.05 if robust else 0.0)
