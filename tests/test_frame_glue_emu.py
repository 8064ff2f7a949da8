"""ComputeStereoFromRGBD / UnprojectStereo device glue (emulator) vs the oracle, bit-exact fp32."""
import numpy as np
from scenes import CAM
from sg_slam_amd import frame as fr
from sg_slam_amd.capi import KP_DTYPE


def run_glue(lib, oracle, S, to_dev=lambda a: a, to_host=lambda a: a):
    g, d, T = S.frame(4)
    d = d.copy(); d[100:140, 200:260] = 0                       # a hole with no depth
    rng = np.random.RandomState(1)
    d = (d.astype('i4') + rng.randint(-300, 300, d.shape)).clip(0, 65535).astype(np.uint16)
    k, _ = oracle.orb_extract(g)
    cap = 1024; n = len(k)
    keys = np.zeros((1, cap), KP_DTYPE); keys[0, :n] = k
    cnt = np.array([n], 'i4')
    ur = np.zeros((1, cap), 'f4'); z = np.zeros((1, cap), 'f4')
    dk, dn, dd, dur, dz = map(to_dev, (keys, cnt, d[None], ur, z))
    fr.stereo_from_rgbd_batch(lib, 1, cap, dk, dn, dd, 640, 480, CAM['depth_factor'], CAM['bf'], dur, dz)
    eur, ez = oracle.compute_stereo_from_rgbd(k, d, CAM['bf'], CAM['depth_factor'])
    ur, z = to_host(dur), to_host(dz)
    assert (ur[0, :n] == eur).all() and (z[0, :n] == ez).all() and (ez < 0).any() and (ur[0, n:] == -1).all()
    th = 0.3
    T32 = T.astype('f4'); T32[:3, :3] = (np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]) @ T[:3, :3]).astype('f4')
    xw = np.zeros((1, cap, 3), 'f4'); has = np.zeros((1, cap), np.uint8)
    dT, dxw, dhas = map(to_dev, (T32.reshape(1, 16).copy(), xw, has))
    fr.unproject_batch(lib, 1, cap, dk, dn, dz, dT, CAM, dxw, dhas)
    exw, ehas = oracle.unproject_stereo(k, ez, T32, CAM)
    xw, has = to_host(dxw), to_host(dhas)
    assert (has[0, :n] == ehas).all() and (xw[0, :n][ehas > 0] == exw[ehas > 0]).all() and (has[0, n:] == 0).all()


def test_glue_emu(emu, oracle, stream_frames):
    run_glue(emu, oracle, stream_frames)
