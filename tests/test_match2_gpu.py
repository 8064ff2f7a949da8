"""LocalMapping matcher gates (tier N2) on the device against the oracle."""
import pytest
import match2_cases as mc

pytestmark = pytest.mark.gpu


def test_hamming_matrix_gpu(gpulib, oracle):
    mc.check_hamming(gpulib, oracle)


def test_search_for_triangulation_gpu(gpulib, oracle):
    mc.check_triangulation(gpulib, oracle, n_cases=8)


def test_search_by_bow_gpu(gpulib, oracle):
    mc.check_bow(gpulib, oracle, n_cases=6)


def test_search_by_bow_kf_gpu(gpulib, oracle):
    mc.check_bow_kf(gpulib, oracle, n_cases=6)


def test_fuse_search_gpu(gpulib, oracle):
    mc.check_fuse(gpulib, oracle, n_cases=6)


def test_project_keyframe_gpu(gpulib, oracle):
    mc.check_project_kf(gpulib, oracle, n_cases=5)


def test_fuse_search_sim3_gpu(gpulib, oracle):
    mc.check_fuse_sim3(gpulib, oracle, n_cases=6)


def test_project_sim3_gpu(gpulib, oracle):
    mc.check_project_sim3(gpulib, oracle, n_cases=6)


def test_search_by_sim3_gpu(gpulib, oracle):
    mc.check_search_by_sim3(gpulib, oracle, n_cases=6)


def test_search_for_initialization_gpu(gpulib, oracle):
    mc.check_search_for_initialization(gpulib, oracle, n_cases=4)
