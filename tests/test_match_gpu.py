"""GPU parity (MI355X) for the matcher and frame glue: identical match pairs vs the oracle."""
import numpy as np
import pytest
from scenes import make_pair, CAM
from sg_slam_amd.matcher import ORBmatcher

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('t,obs_mode,th,seed', [(0, 'mixed', 15, 0), (3, 'zero', 15, 1), (10, 'all', 15, 2), (20, 'mixed', 30, 3), (33, 'all', 7, 4), (60, 'mixed', 15, 5)])
def test_search_by_projection_frame(gpulib, oracle, stream_frames, t, obs_mode, th, seed):
    cur, last = make_pair(oracle, stream_frames, t, seed=seed, obs_mode=obs_mode, pose_noise=0.002 if seed % 2 else 0.0)
    sf = oracle.orb_params()['scale']
    exp_match, exp_n = oracle.search_by_projection_frame(cur, last, CAM, sf, th=th, mono=False, check_ori=True)
    n = ORBmatcher(0.9, True, lib=gpulib).SearchByProjection(cur, last, th, False, CAM, sf)
    assert exp_n > 100 and n == exp_n and (cur['match'] == exp_match).all()


def test_lock_rule_adversarial(gpulib, oracle):
    from test_match_emu import test_lock_rule_adversarial as body
    body(gpulib, oracle)


def test_modes(gpulib, oracle, stream_frames):
    from test_match_emu import test_no_orientation_check_and_mono as body
    body(gpulib, oracle, stream_frames)


def test_glue_gpu(gpulib, oracle, stream_frames):
    import torch
    from test_frame_glue_emu import run_glue
    run_glue(gpulib, oracle, stream_frames,
             to_dev=lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else (a.view(np.int16) if a.dtype == np.uint16 else a)).cuda(),
             to_host=lambda t: t.cpu().numpy())


def test_map_glue_gpu(gpulib, oracle, stream_frames):
    import torch
    from test_frame_glue_emu import run_map_glue
    run_map_glue(gpulib, oracle, stream_frames,
                 to_dev=lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else (a.view(np.int16) if a.dtype == np.uint16 else a)).cuda(),
                 to_host=lambda t: t.cpu().numpy())


def test_gray_gpu(gpulib):
    import torch
    from test_frame_glue_emu import run_gray
    run_gray(gpulib, to_dev=lambda a: torch.from_numpy(a).cuda(), to_host=lambda t: t.cpu().numpy())
