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


def test_lk_beside_bf16_matrix_products_gpu(gpulib, gpulib_taps):
    """Round 6: beside a co-resident wave that issues bf16 matrix products back to back (the detector's k_hrb blocks do, on their own stream), compiler-generated PACKED fp32
    instructions returned wrong values in lanes 48-63 — the LK tracker's step arithmetic had been paired into v_pk_fma_f32 / v_pk_mul_f32 by the SLP vectoriser and 50-100 of
    2 000 keypoints per run came out different (20 of 20 runs; profiles/r6_lk_priority_diagnosis.md).  The library is built with -fno-slp-vectorize; this runs the PRODUCT
    library's tracker beside the tap library's co-runner (sgx_debug_corun_bf16: 1024 workgroups of nothing but v_mfma_f32_32x32x16_bf16) and demands the quiet run's bits."""
    import ctypes as C
    import torch
    from sg_slam_amd import synth
    from sg_slam_amd.flow import OpticalFlowLK
    from sg_slam_amd.orb import ORBextractor
    S = 2; gen = synth.PlaneStream(seed=1234); offs = [3, 57]
    f0 = _xp(np.stack([gen.frame(o + 1)[0] for o in offs])); f1 = _xp(np.stack([gen.frame(o + 2)[0] for o in offs]))
    ex = ORBextractor(nfeatures=1000, width=640, height=480, max_batch=S, lib=gpulib); cap = ex.capacity
    keys = torch.zeros((S, cap, 28), dtype=torch.uint8, device='cuda'); desc = torch.zeros((S, cap, 32), dtype=torch.uint8, device='cuda'); n = torch.zeros(S, dtype=torch.int32, device='cuda')
    ex.extract_batch_dev(f1, 640, S, keys, desc, n); torch.cuda.synchronize(); nn = n.cpu().numpy()
    assert nn.min() > 500
    fl = OpticalFlowLK(width=640, height=480, max_batch=S, lib=gpulib); sV, sC = torch.cuda.Stream(), torch.cuda.Stream()
    xy = torch.zeros((S, cap, 2), dtype=torch.float32, device='cuda'); status = torch.zeros((S, cap), dtype=torch.uint8, device='cuda')

    def run(co):
        torch.cuda.synchronize()
        if co: gpulib_taps.check(gpulib_taps.tap('sgx_debug_corun_bf16')(1024, 2000, 3, C.c_void_p(sC.cuda_stream)))
        fl.reset(); fl.lk_batch_dev(f0, 640, S, None, None, cap, None, None, stream=sV.cuda_stream); fl.lk_batch_dev(f1, 640, S, keys, n, cap, xy, status, stream=sV.cuda_stream)
        torch.cuda.synchronize()
        return [np.concatenate([xy[s, :nn[s]].cpu().numpy().view(np.uint32), status[s, :nn[s], None].cpu().numpy().astype(np.uint32)], 1) for s in range(S)]
    ref = run(False)
    for r in range(10):
        out = run(True)
        for s in range(S):
            differ = int((out[s] != ref[s]).any(1).sum())
            assert differ == 0, 'repetition %d stream %d: %d of %d keypoints differ from the quiet run' % (r, s, differ, nn[s])
    fl.close(); ex.close()
