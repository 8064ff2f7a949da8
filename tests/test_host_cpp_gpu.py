"""GPU: the C++ host mirror classes (sg_slam_amd/host/sgx_host.hpp) driven by example_track.cpp — a
reference-style Frame -> SearchByProjection -> PoseOptimization sequence — against the oracle."""
import os
import subprocess
import numpy as np
import pytest
from scenes import CAM

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_example_matches_oracle(gpulib, oracle, stream_frames, tmp_path):
    exe = os.path.join(ROOT, 'sg_slam_amd', 'host', 'example_track')
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    g0, _, _ = stream_frames.frame(10); g1, _, _ = stream_frames.frame(11)
    f0 = tmp_path / 'f0.raw'; f1 = tmp_path / 'f1.raw'
    g0.tofile(f0); g1.tofile(f1)
    out = subprocess.check_output([exe, str(f0), str(f1)], text=True).splitlines()
    vals = out[0].split()
    n0, n1, nm, ninl = int(vals[1]), int(vals[3]), int(vals[5]), int(vals[7])
    T = np.array([float(v) for v in out[1].split()[1:]], 'f4').reshape(4, 4)
    # oracle, chained the same way (constant depth 2 m, identity last pose)
    k0, d0 = oracle.orb_extract(g0); k1, d1 = oracle.orb_extract(g1)
    z = np.float32(2.0)
    ur0 = (k0['x'] - np.float32(CAM['bf']) / z).astype('f4'); ur1 = (k1['x'] - np.float32(CAM['bf']) / z).astype('f4')
    xw = np.stack([(k0['x'] - np.float32(CAM['cx'])) * z * (np.float32(1) / np.float32(CAM['fx'])),
                   (k0['y'] - np.float32(CAM['cy'])) * z * (np.float32(1) / np.float32(CAM['fy'])), np.full(len(k0), z)], 1).astype('f4')
    I = np.eye(4, dtype='f4')
    last = dict(keys=k0, has_mp=np.ones(len(k0), np.uint8), outlier=np.zeros(len(k0), np.uint8), xw=xw, obs=np.zeros(len(k0), 'i4'), mpdesc=d0, Tcw=I)
    cur = dict(keys=k1, desc=d1, uright=ur1, Tcw=I)
    sf = oracle.orb_params()['scale']; is2 = oracle.orb_params()['inv_sigma2']
    m, en = oracle.search_by_projection_frame(cur, last, CAM, sf, th=15)
    if en < 20:
        m, en = oracle.search_by_projection_frame(cur, last, CAM, sf, th=30)
    fr = dict(keys=k1, uright=ur1, has_mp=(m >= 0).astype(np.uint8), Tcw=I, xw=np.where((m >= 0)[:, None], xw[np.maximum(m, 0)], 0).astype('f4'))
    einl, eT, _ = oracle.pose_optimization(fr, CAM, is2)
    assert (n0, n1, nm, ninl) == (len(k0), len(k1), en, einl)
    assert np.abs(T - eT).max() <= 1e-5 * max(1.0, np.abs(eT).max())
