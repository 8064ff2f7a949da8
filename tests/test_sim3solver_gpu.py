"""Sim3Solver on the device against the oracle (device libm differs from the host's in the last bit of hypot / sin / cos / atan2: models within 2e-5, inlier sets within one pair)."""
import pytest
import sim3solver_cases as sc

pytestmark = pytest.mark.gpu


def test_sim3_solver_gpu(gpulib, oracle):
    sc.check_solver(gpulib, oracle, n_cases=5, exact=False)
