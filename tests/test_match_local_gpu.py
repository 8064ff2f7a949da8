"""GPU (MI355X): local-map matcher parity."""
import pytest
from test_match_local import run_local

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('t,seed,th', [(5, 0, 3.0), (12, 1, 3.0), (30, 2, 5.0), (44, 3, 1.0), (70, 4, 3.0)])
def test_local_gpu(gpulib, oracle, stream_frames, t, seed, th):
    run_local(gpulib, oracle, stream_frames, t, seed, th)


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_local_crowded_gpu(gpulib, oracle, seed):
    from test_match_local import run_crowded
    run_crowded(gpulib, oracle, seed)
