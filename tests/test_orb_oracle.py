"""Oracle self-checks for the ORB extractor: constants the SURVEY derives from the reference,
and analytic known-answer tests (the reference ships no golden vectors for this path)."""
import numpy as np
import pytest
from sg_slam_amd import synth


def test_constructor_tables(oracle):
    # quotas, umax and level sizes for TUM3.yaml (1000 feats, 1.2, 8 levels, 640x480)
    p = oracle.orb_params(1000, 1.2, 8)
    assert list(p['per_level']) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(p['umax']) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert oracle.level_sizes(640, 480) == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    assert np.float32(p['scale'][1]) == np.float32(1.2)
    # size field = (int)(31*scale): Appendix B
    assert [int(31 * s) for s in p['scale']] == [31, 37, 44, 53, 64, 77, 92, 111]


def test_fast_handmade_patterns(oracle):
    # a single bright pixel on a flat background: all 16 ring pixels darker by 100 -> corner, score 99
    img = np.full((15, 15), 50, np.uint8); img[7, 7] = 150
    x, y, s = oracle.fast(img, 20)
    assert list(zip(x, y, s)) == [(7, 7, 99)]
    # a 9-pixel contiguous brighter arc (ring indices 0..8) is a corner; 8 is not
    base = np.full((15, 15), 100, np.uint8)
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    im9 = base.copy()
    for dx, dy in ring[:9]:
        im9[7 + dy, 7 + dx] = 160
    x, y, s = oracle.fast(im9, 20, nonmax=False)
    assert (7, 7) in list(zip(x, y))
    k = list(zip(x, y)).index((7, 7))
    assert s[k] == 59                     # min |diff| on the arc = 60 -> score 60-1
    im8 = base.copy()
    for dx, dy in ring[:8]:
        im8[7 + dy, 7 + dx] = 160
    x, y, s = oracle.fast(im8, 20, nonmax=False)
    assert (7, 7) not in list(zip(x, y))
    # threshold is strict: diff == t is not a corner
    im_eq = base.copy()
    for dx, dy in ring[:9]:
        im_eq[7 + dy, 7 + dx] = 120
    x, y, s = oracle.fast(im_eq, 20, nonmax=False)
    assert (7, 7) not in list(zip(x, y))
    # NMS is strict: two adjacent corners with equal scores suppress each other
    two = np.full((15, 16), 50, np.uint8); two[7, 7] = 150; two[7, 8] = 150
    x, y, s = oracle.fast(two, 20, nonmax=True)
    assert (7, 7) not in list(zip(x, y)) and (8, 7) not in list(zip(x, y))


def test_fast_score_is_threshold_independent(oracle):
    # the property the GPU score-map formulation relies on
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (40, 40)).astype(np.uint8)
    x7, y7, s7 = oracle.fast(img, 7, nonmax=False)
    x20, y20, s20 = oracle.fast(img, 20, nonmax=False)
    d7 = {(a, b): c for a, b, c in zip(x7, y7, s7)}
    d20 = {(a, b): c for a, b, c in zip(x20, y20, s20)}
    assert len(d20) > 0
    assert all(d7[k] == v for k, v in d20.items())
    assert {k for k, v in d7.items() if v >= 20} == set(d20)


def test_descriptor_constant_image_is_zero(oracle):
    blur = np.full((80, 80), 77, np.uint8)
    d = oracle.descriptor(blur, 40, 40, 33.0)
    assert (d == 0).all()                 # t0 < t1 never holds


def test_gaussian_and_resize_identities(oracle):
    flat = np.full((50, 60), 131, np.uint8)
    assert (oracle.gaussian7(flat) == 131).all()        # taps sum to exactly 256
    assert (oracle.resize_linear(flat, 50, 42) == 131).all()
    # impulse response of the blur = outer product of the 8.8 taps, rounded
    imp = np.zeros((21, 21), np.uint8); imp[10, 10] = 255
    g = oracle.gaussian7(imp)
    taps = np.array([18, 34, 48, 56, 48, 34, 18])
    exp = ((np.outer(taps, taps) * 255 + 32768) >> 16).astype(np.uint8)
    assert (g[7:14, 7:14] == exp).all()
    assert g.sum() == exp.sum()


def test_fast_atan2_quadrants(oracle):
    assert abs(oracle.fast_atan2(0, 1) - 0) < 1e-3
    assert abs(oracle.fast_atan2(1, 0) - 90) < 1e-3
    assert abs(oracle.fast_atan2(0, -1) - 180) < 1e-3
    assert abs(oracle.fast_atan2(-1, 0) - 270) < 1e-3
    for a in np.linspace(0.5, 359.5, 97):
        r = np.radians(a)
        assert abs(oracle.fast_atan2(float(np.sin(r)), float(np.cos(r))) - a) < 0.02   # cv::fastAtan2 accuracy ~0.01 deg


def test_degenerate_images(oracle):
    k, d = oracle.orb_extract(synth.constant_image())
    assert len(k) == 0 and d.shape == (0, 32)
    k, d, _, nc = oracle.orb_extract(synth.low_contrast_image(), want_pyr=True)
    assert len(k) > 100                   # every cell took the minThFAST branch
    assert (k['response'] >= 7).all() and (k['response'] < 20).mean() > 0.5    # mostly fallback-threshold corners


def test_extract_invariants(oracle, stream_frames):
    g, _, _ = stream_frames.frame(0)
    k, d = oracle.orb_extract(g)
    p = oracle.orb_params()
    assert 900 <= len(k) <= 1024 and d.shape == (len(k), 32)
    assert (np.diff(k['octave']) >= 0).all()                 # grouped by level, ascending
    cnt = np.bincount(k['octave'], minlength=8)
    assert (cnt <= p['per_level'] + 3).all()
    assert (k['angle'] >= 0).all() and (k['angle'] < 360).all()
    assert (k['class_id'] == -1).all()
    sizes = oracle.level_sizes(640, 480)
    for l in range(8):
        m = k['octave'] == l
        x = k['x'][m] / p['scale'][l]; y = k['y'][m] / p['scale'][l]
        assert (x >= 19 - 1e-3).all() and (x <= sizes[l][0] - 20 + 1e-3).all()
        assert (y >= 19 - 1e-3).all() and (y <= sizes[l][1] - 20 + 1e-3).all()
