"""Batched MapPoint post-steps on the device against the oracle."""
import pytest
import mappoint_cases as mpc

pytestmark = pytest.mark.gpu


def test_mappoint_post_steps_gpu(gpulib, oracle):
    mpc.check_mappoint(gpulib, oracle, n_cases=3)
