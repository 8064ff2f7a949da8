"""Shared test scenes: consecutive synthetic RGB-D frames turned into the flattened Frame arrays the
matcher / pose optimiser consume (built with the oracle's extractor + stereo helpers)."""
import numpy as np
from sg_slam_amd import synth

CAM = dict(synth.TUM3)


def make_pair(orc, S, t, seed=0, obs_mode='mixed', pose_noise=0.0):
    """last = frame t, cur = frame t+1.  Every last-frame keypoint with depth carries a map point
    (UnprojectStereo with the true pose).  obs_mode: 'zero' (visual-odometry points, no locks),
    'mixed' (half observed -> exercises the greedy lock rule), 'all'."""
    rng = np.random.RandomState(seed)
    gl, dl, Tl = S.frame(t)
    gc, dc, Tc = S.frame(t + 1)
    kl, desl = orc.orb_extract(gl)
    kc, desc = orc.orb_extract(gc)
    url, zl = orc.compute_stereo_from_rgbd(kl, dl, CAM['bf'], CAM['depth_factor'])
    urc, zc = orc.compute_stereo_from_rgbd(kc, dc, CAM['bf'], CAM['depth_factor'])
    Tl32 = Tl.astype('f4')
    xw, has = orc.unproject_stereo(kl, zl, Tl32, CAM)
    n = len(kl)
    if obs_mode == 'zero':
        obs = np.zeros(n, 'i4')
    elif obs_mode == 'all':
        obs = np.full(n, 3, 'i4')
    else:
        obs = (rng.rand(n) < 0.5).astype('i4') * rng.randint(1, 6, n)
    outlier = (rng.rand(n) < 0.03).astype(np.uint8)
    mpdesc = desl.copy()
    flip = rng.rand(n) < 0.3                       # distinctive descriptors differ a little from the keypoint's own
    for i in np.nonzero(flip)[0]:
        b = rng.randint(0, 256, 6)
        mpdesc[i, b // 8] ^= (1 << (b % 8)).astype(np.uint8)
    Tc32 = Tc.astype('f4')
    if pose_noise:
        Tc32 = Tc32.copy(); Tc32[:3, 3] += rng.randn(3).astype('f4') * pose_noise
    last = dict(keys=kl, desc=desl, uright=url, zdepth=zl, Tcw=Tl32, has_mp=has, outlier=outlier, xw=xw, obs=obs, mpdesc=mpdesc)
    cur = dict(keys=kc, desc=desc, uright=urc, zdepth=zc, Tcw=Tc32)
    return cur, last
