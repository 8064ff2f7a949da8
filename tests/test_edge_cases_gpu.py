"""GPU (MI355X): the degenerate-input cases of test_edge_cases.py through the device library."""
import pytest
from test_edge_cases import run_edge_cases

pytestmark = pytest.mark.gpu


def test_edge_cases_gpu(gpulib, oracle, stream_frames):
    run_edge_cases(gpulib, oracle, stream_frames)
