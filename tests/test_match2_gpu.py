"""LocalMapping matcher gates (tier N2) on the device against the oracle."""
import pytest
import match2_cases as mc

pytestmark = pytest.mark.gpu


def test_hamming_matrix_gpu(gpulib, oracle):
    mc.check_hamming(gpulib, oracle)


def test_search_for_triangulation_gpu(gpulib, oracle):
    mc.check_triangulation(gpulib, oracle, n_cases=8)
