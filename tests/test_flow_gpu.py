"""Mask inputs (tier N1) on the device: pyramidal LK flow bit-identical to the oracle's exact-accumulation variant, RANSAC fundamental matrix with
the oracle's sampling trajectory — through the C-ABI of libsgx.so (HIP, gfx950)."""
import numpy as np
import pytest
import flow_cases as fc

pytestmark = pytest.mark.gpu


def _xp(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_pyramid_gpu(gpulib_taps, oracle):
    gpulib = gpulib_taps          # reads the LK pyramid levels back (include/sgx_debug.h)
    fc.check_pyramid(gpulib, oracle)


@pytest.mark.parametrize('t', [11, 37, 64])
def test_lk_pair_gpu(gpulib, oracle, t):
    fc.check_lk_pair(gpulib, oracle, t)


def test_lk_textureless_gpu(gpulib, oracle):
    fc.check_lk_textureless(gpulib, oracle)


def test_lk_stream_gpu(gpulib, oracle):
    fc.check_lk_stream(gpulib, oracle, _xp, S=3, T=4)


def test_lk_noise_gpu(gpulib, oracle):
    """pure noise vs shifted noise + unrelated image: many iterations, oscillation / out-of-image exits, re-staged LDS tiles"""
    from sg_slam_amd.flow import OpticalFlowLK
    rng = np.random.RandomState(7)
    a = rng.randint(0, 256, (480, 640)).astype(np.uint8)
    b = np.roll(a, (3, -5), (0, 1)); c = rng.randint(0, 256, (480, 640)).astype(np.uint8)
    pts = np.c_[rng.uniform(0, 640, 600), rng.uniform(0, 480, 600)].astype('f4')
    fl = OpticalFlowLK(lib=gpulib)
    for J in (b, c):
        got, st = fl(a, J, pts)
        ref, rst, it = oracle.lk_pyr(a, J, pts, want_iters=True)
        assert (st == rst).all() and (got.view(np.uint32) == ref.view(np.uint32)).all()
    assert it.max() >= 10
    fl.close()


def test_ransac_host_gpu(gpulib, oracle):
    fc.check_ransac_host(gpulib, oracle, exact_lmeds=False)


def test_ransac_batch_gpu(gpulib, oracle):
    fc.check_ransac_batch(gpulib, oracle, _xp)


def test_ransac_degenerate_gpu(gpulib, oracle):
    """all pairs on one line: every 7-subset fails checkSubset; getSubset gives up after 10000 attempts -> empty Mat, and the kernel terminates"""
    from sg_slam_amd.flow import find_fundamental_mat
    x = np.arange(40, dtype='f4')
    p1 = np.stack([x * 10, x * 5 + 3], 1); p2 = p1 + np.float32(2.0)
    ok, F, st = find_fundamental_mat(p1, p2, lib=gpulib)
    rok, _, _, _ = oracle.find_fundamental_ransac(p1, p2)
    assert ok == rok == 0 and (F == 0).all()
