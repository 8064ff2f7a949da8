"""TUM I/O + trajectory evaluation (sg_slam_amd/tum.py): association parsing (rgbd_tum.cc:258-283), trajectory line format (System.cc:444-452), ATE."""
import os
import numpy as np
from sg_slam_amd import tum, synth


def test_associations_and_frames_roundtrip(tmp_path):
    gen = synth.LayeredStream(seed=5)
    fr = [gen.frame(t) for t in range(3)]
    stamps = [1305031102.175304 + 0.033 * i for i in range(3)]
    tum.write_sequence(str(tmp_path), stamps, [f[0] for f in fr], [f[1] for f in fr])
    with open(os.path.join(tmp_path, 'associations.txt'), 'a') as f:
        f.write('\n\n')                                          # trailing empty lines are skipped like `if(!s.empty())`
    st, rgb, dep = tum.load_associations(os.path.join(tmp_path, 'associations.txt'))
    assert len(st) == 3 and np.allclose(st, stamps, atol=1e-6)
    for i in range(3):
        bgr, d = tum.load_frame(str(tmp_path), rgb[i], dep[i])
        assert bgr.shape == (480, 640, 3) and (bgr[:, :, 0] == fr[i][0]).all() and (bgr[:, :, 2] == fr[i][0]).all()
        assert d.dtype == np.uint16 and (d == fr[i][1]).all()


def test_trajectory_format_and_quaternion(tmp_path):
    gen = synth.PlaneStream(seed=1)
    T = [gen.Tcw(t).astype('f4') for t in range(5)]
    lines = tum.trajectory_lines([0.5 + i for i in range(5)], T)
    assert lines[0].replace('-0.000000000', '0.000000000') == '0.500000 0.000000000 0.000000000 0.000000000 0.000000000 0.000000000 0.000000000 1.000000000'      # first pose = origin (C++ prints -0 the same way)
    for l in lines:
        tok = l.split()
        assert len(tok) == 8 and len(tok[0].split('.')[1]) == 6 and all(len(x.split('.')[1]) == 9 for x in tok[1:])
        q = np.array([float(x) for x in tok[4:]]); assert abs(np.linalg.norm(q) - 1) < 1e-6
    # Shepperd branches against a rotation with negative trace
    R = np.diag([1.0, -1.0, -1.0]); q = tum.quaternion_from_rotation(R)
    assert np.allclose(np.abs(q), [1, 0, 0, 0])
    p = os.path.join(tmp_path, 'traj.txt'); tum.save_trajectory_tum(p, [0.5 + i for i in range(5)], T)
    st, xyz, qq = tum.load_trajectory_tum(p)
    assert len(st) == 5 and xyz.shape == (5, 3) and qq.shape == (5, 4)


def test_ate_is_invariant_to_rigid_motion_and_measures_noise():
    rng = np.random.RandomState(0)
    ref = np.cumsum(rng.normal(0, 0.05, (60, 3)), 0)
    a = 0.7; R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    est = (R @ ref.T).T + np.array([3.0, -1.0, 0.5])
    assert tum.ate_rmse(est, ref) < 1e-9
    noise = rng.normal(0, 0.01, ref.shape)
    r = tum.ate_rmse(est + noise, ref)
    assert 0.012 < r < 0.022                                     # sqrt(3) * 0.01, minus what the alignment absorbs
    assert tum.associate([0.0, 0.1, 0.2], [0.005, 0.2, 0.1001]) == [(0, 0), (1, 2), (2, 1)]
