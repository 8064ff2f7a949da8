"""GPU (MI355X): the batched tracking harness stage-by-stage against the chained oracle."""
import pytest
from test_tracker_emu import run_tracker

pytestmark = pytest.mark.gpu


def test_tracker_two_streams_gpu(gpulib, oracle):
    run_tracker(gpulib, oracle, 'torch')


def test_tracker_mask_gpu(gpulib):
    from test_tracker_emu import run_tracker_mask
    run_tracker_mask(gpulib, 'torch')
