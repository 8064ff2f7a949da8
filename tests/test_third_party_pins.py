"""The oracle against outputs of the REAL third-party libraries (OpenCV 3.4.x, ncnn), primitive by primitive.  The fixtures are produced by tools/pin_third_party.py
on a machine that has those libraries (this image and /root/reference have neither) and committed under tests/golden/third_party/; until then every test here is
skipped and oracle/ stays "parity unpinned" at the third-party boundaries."""
import os
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TP = os.path.join(ROOT, 'tests', 'golden', 'third_party')


def _load(name):
    p = os.path.join(TP, name)
    if not os.path.exists(p):
        pytest.skip(f'{name} absent: run tools/pin_third_party.py where the real library is installed')
    return np.load(p, allow_pickle=False)


def test_cv_resize_chain(oracle):
    z = _load('opencv.npz')
    lv = z['resize_src']
    for i in range(1, 8):
        ref = z[f'resize_l{i}']
        lv = oracle.resize_linear(lv, ref.shape[1], ref.shape[0])
        assert (lv == ref).all(), i


def test_cv_fast(oracle):
    z = _load('opencv.npz')
    for t in (20, 7):
        for name, img in (('full', z['fast_src']), ('crop', np.ascontiguousarray(z['fast_src'][100:136, 200:236]))):
            ref = z[f'fast_{name}_t{t}']
            x, y, r = oracle.fast(img, t, nonmax=True)
            got = np.stack([x, y, r], 1).astype(np.float32)
            assert got.shape == ref.shape and (got == ref).all(), (name, t)


def test_cv_gaussian_blur(oracle):
    z = _load('opencv.npz')
    assert (oracle.gaussian7(z['blur_src']) == z['blur']).all()


def test_cv_fast_atan2(oracle):
    z = _load('opencv.npz')
    got = np.array([oracle.fast_atan2(float(y), float(x)) for y, x in z['atan2_yx']], np.float32)
    assert (got.view(np.uint32) == z['atan2'].view(np.uint32)).all()


def test_cv_cvtcolor(emu):
    """the product's BGR2GRAY kernel logic (emulator build) against cv::cvtColor"""
    from sg_slam_amd.capi import _vp
    z = _load('opencv.npz')
    src = z['cvt_src']; h, w = src.shape[:2]
    for blue_first, key in ((0, 'cvt_rgb2gray'), (1, 'cvt_bgr2gray')):
        out = np.zeros((1, h, w), np.uint8)
        emu.check(emu.dll.sgx_frame_gray_from_color_batch_dev(1, w, h, _vp(np.ascontiguousarray(src[None])), w * 3, 3, blue_first, _vp(out), w, None))
        assert (out[0] == z[key]).all(), key


def test_cv_pyrdown(oracle):
    z = _load('opencv.npz')
    p1 = oracle.pyr_down(z['lk_prev'])
    assert (p1 == z['pyrdown_1']).all() and (oracle.pyr_down(p1) == z['pyrdown_2']).all()


def test_cv_lk(oracle):
    """OpenCV's own float accumulation order (acc_mode 0) is what a given build runs: positions to 0.01 px (the build's SIMD path decides the last bits), status exactly"""
    z = _load('opencv.npz')
    nxt, st = oracle.lk_pyr(z['lk_cur'], z['lk_prev'], z['lk_pts'], acc_mode=0)
    assert (st == z['lk_status']).all()
    ok = st > 0
    assert np.abs(nxt[ok] - z['lk_next'][ok]).max() < 0.01


def test_cv_find_fundamental_mat(oracle):
    z = _load('opencv.npz')
    ok, F, mask, stats = oracle.find_fundamental_ransac(z['fm_p1'], z['fm_p2'])
    assert ok == 1 and (mask.astype(bool) == z['fm_mask'].astype(bool)).all()
    Fr = z['fm_F']
    assert np.abs(F / F[2, 2] - Fr / Fr[2, 2]).max() < 1e-6 * np.abs(Fr / Fr[2, 2]).max()


def test_ncnn_blobs():
    """numpy port of the ncnn layers (oracle/detector_oracle.py) against ncnn itself on the shipped graph: fp32 drift bound per blob; the pre-processing exactly"""
    z = _load('ncnn.npz')
    from oracle import detector_oracle as D
    from sg_slam_amd import synth
    if 'synthetic' not in str(z['info'][0]):
        pytest.skip('fixture was made with real weights: pass the same .bin to the oracle to use it')
    param = os.path.join(ROOT, 'tests', 'golden', 'mobilenetv3_ssdlite_voc.param')
    layers = D.parse_param(param); W, _ = D.synth_weights(layers)
    x = D.preprocess(z['input_bgr'])
    assert (x == z['input_blob'].reshape(x.shape)).all()
    _, blobs = D.forward(layers, W, x)
    for k in z.files:
        if not k.startswith('blob_') or k[5:] not in blobs: continue
        ref = z[k].reshape(-1).astype(np.float64); got = np.asarray(blobs[k[5:]], np.float64).reshape(-1)
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), k
