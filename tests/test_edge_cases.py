"""Degenerate inputs through the host mirrors (emulator here, device twin in test_edge_cases_gpu.py): empty frames, empty local maps, frames without
map points — the reference returns 0 / leaves outputs untouched in these cases (ORBmatcher.cc:1332-1472 loops do not execute, Optimizer.cc:364-365)."""
import numpy as np
from scenes import make_pair, make_local_map, make_pose_problem, CAM
from sg_slam_amd.capi import KP_DTYPE
from sg_slam_amd.matcher import ORBmatcher
from sg_slam_amd.optimizer import Optimizer


def run_edge_cases(lib, oracle, S):
    sf = oracle.orb_params()['scale']; is2 = oracle.orb_params()['inv_sigma2']
    cur, last = make_pair(oracle, S, 6, 1)
    empty = dict(keys=np.zeros(0, KP_DTYPE), desc=np.zeros((0, 32), np.uint8), uright=np.zeros(0, 'f4'), Tcw=cur['Tcw'])
    # current frame without keypoints
    n = ORBmatcher(0.9, True, lib=lib).SearchByProjection(empty, last, 15, False, CAM, sf)
    assert n == 0 and len(empty['match']) == 0
    # last frame without keypoints / without any map point
    last0 = {k: (v[:0] if hasattr(v, '__len__') and k != 'Tcw' else v) for k, v in last.items()}
    c2 = dict(cur)
    assert ORBmatcher(0.9, True, lib=lib).SearchByProjection(c2, last0, 15, False, CAM, sf) == 0 and (c2['match'] == -1).all()
    last1 = dict(last); last1['has_mp'] = np.zeros_like(last['has_mp'])
    c3 = dict(cur)
    assert ORBmatcher(0.9, True, lib=lib).SearchByProjection(c3, last1, 15, False, CAM, sf) == 0 and (c3['match'] == -1).all()
    _, en = oracle.search_by_projection_frame(dict(cur), last1, CAM, sf, th=15)
    assert en == 0
    # local map: every point skipped, and an empty local map
    F, lm = make_local_map(oracle, S, 9, seed=2)
    lm_skip = dict(lm); lm_skip['skip'] = np.ones_like(lm['skip'])
    F2 = dict(F)
    assert ORBmatcher(0.8, True, lib=lib).SearchByProjectionLocal(F2, lm_skip, 3.0, CAM, sf) == 0 and (F2['match_local'] == -1).all() and (lm_skip['in_view'] == 0).all()
    lm0 = {k: v[:0] for k, v in lm.items()}
    F3 = dict(F)
    assert ORBmatcher(0.8, True, lib=lib).SearchByProjectionLocal(F3, lm0, 3.0, CAM, sf) == 0 and (F3['match_local'] == -1).all()
    # pose optimisation: no keypoints, keypoints without map points, exactly 3 correspondences (the minimum the reference optimises, Optimizer.cc:364)
    fr, _, _ = make_pose_problem(oracle, n=50, seed=3)
    f0 = {k: (v[:0] if k != 'Tcw' else v) for k, v in fr.items()}
    assert Optimizer.PoseOptimization(f0, CAM, is2, lib=lib) == 0 and (f0['Tcw'] == fr['Tcw']).all()
    f1 = dict(fr); f1['has_mp'] = np.zeros_like(fr['has_mp'])
    assert Optimizer.PoseOptimization(f1, CAM, is2, lib=lib) == 0 and (f1['Tcw'] == fr['Tcw']).all() and (f1['outlier'] == 0).all()
    f3 = dict(fr); h = np.zeros_like(fr['has_mp']); h[[4, 17, 33]] = 1; f3['has_mp'] = h
    e3 = dict(f3); en3, eT3, eo3 = oracle.pose_optimization(e3, CAM, is2)
    g3 = Optimizer.PoseOptimization(f3, CAM, is2, lib=lib)
    assert g3 == en3 and (f3['outlier'] == eo3).all() and np.abs(f3['Tcw'] - eT3).max() <= 1e-5 * max(1.0, np.abs(eT3).max())

    # argument validation of the C ABI: bad arguments are reported (SGX_ERR_INVALID = -1), never dereferenced
    import ctypes as C
    d = lib.dll
    one = np.zeros(64, 'f4'); i1 = np.zeros(4, 'i4')
    assert d.sgx_match_project_local(-1, None, None, None, one.ctypes.data, None, 0, None, None, None, None, None, None, None, None, None, 8, 0.18, 3.0, 0.8, 0.5,
                                     i1.ctypes.data, i1.ctypes.data, None) == -1
    assert d.sgx_match_project_local(0, None, None, None, one.ctypes.data, None, 0, None, None, None, None, None, None, None, None, None, 8, 0.18, 3.0, 0.8, 0.5,
                                     None, i1.ctypes.data, None) == -1
    assert d.sgx_pose_optimization(5000, None, None, None, None, None, 8, None, one.ctypes.data, i1.ctypes.data, i1.ctypes.data) == -1
    assert d.sgx_det_detect_batch_dev(None, None, 0, 1, None, None, None, 0, None, None) == -1
    assert d.sgx_det_detect(None, None, 0, 1, None) == -1
    assert d.sgx_local_bundle_adjustment(None, None, None, None, None) == -1
    assert d.sgx_orb_extract_batch_dev(None, None, 640, 1, None, None, None, 1024, None) == -1


def test_edge_cases_emu(emu, oracle, stream_frames):
    run_edge_cases(emu, oracle, stream_frames)
