"""GPU (MI355X): Optimizer::OptimizeSim3 through the C-ABI against the oracle."""
import pytest
import sim3_cases as sc

pytestmark = pytest.mark.gpu


def test_sim3_gpu(gpulib, oracle):
    sc.check_sim3(gpulib, oracle, n_cases=6)
