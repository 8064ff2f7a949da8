"""Kernel LOGIC parity without a GPU: the product's HIP kernel sources compiled with -DSGX_EMU
(tests/emu/libsgx_emu.so, sequential workgroup emulation) against the CPU oracle, bit-exact.
The same assertions run on the real device in tests/test_orb_gpu.py."""
import numpy as np
import pytest
from sg_slam_amd import synth
from sg_slam_amd.orb import ORBextractor


@pytest.fixture(scope='module')
def ex(emu):
    e = ORBextractor(lib=emu, max_batch=2)
    yield e
    e.close()


def _same(ka, da, kb, db):
    return len(ka) == len(kb) and (ka == kb).all() and da.shape == db.shape and (da == db).all()


@pytest.mark.parametrize('t', [0, 1, 7, 23])
def test_extract_matches_oracle(ex, oracle, stream_frames, t):
    g, _, _ = stream_frames.frame(t)
    k, d = ex(g)
    ko, do = oracle.orb_extract(g)
    assert _same(k, d, ko, do)


def test_stage_taps_match_oracle(ex, oracle, stream_frames):
    g, _, _ = stream_frames.frame(5)
    ex(g)
    _, _, pyr, nc = oracle.orb_extract(g, want_pyr=True)
    off = 640 * 480
    for l, (w, h) in enumerate(oracle.level_sizes(640, 480)):
        if l == 0:
            lvl = g
        else:
            lvl = pyr[off:off + w * h].reshape(h, w); off += w * h
            assert (ex.debug_level(0, l) == lvl).all()
        x, y, s = ex.debug_candidates(0, l)
        cx, cy, cr = oracle.level_candidates(lvl)
        assert len(x) == nc[l] == len(cx)
        assert sorted(zip(x, y, s)) == sorted(zip(cx.astype(int), cy.astype(int), cr.astype(int)))


def test_degenerate_images(ex, oracle):
    k, d = ex(synth.constant_image())
    assert len(k) == 0 and d.shape == (0, 32)
    img = synth.low_contrast_image()
    k, d = ex(img)
    ko, do = oracle.orb_extract(img)
    assert _same(k, d, ko, do) and len(k) > 100


def test_random_noise_image(ex, oracle):
    # pure noise: ~every cell saturated with corners, stresses candidate volume and octree phase 2
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (480, 640)).astype(np.uint8)
    k, d = ex(img)
    ko, do = oracle.orb_extract(img)
    assert _same(k, d, ko, do)


def test_batched_call_equals_single(ex, oracle, stream_frames, emu):
    g0, _, _ = stream_frames.frame(2); g1, _, _ = stream_frames.frame(3)
    batch = np.stack([g0, g1])
    cap = ex.capacity
    from sg_slam_amd.capi import KP_DTYPE
    kps = np.zeros((2, cap), KP_DTYPE); desc = np.zeros((2, cap, 32), np.uint8); cnt = np.zeros(2, 'i4')
    ex.extract_batch_dev(batch, 640, 2, kps, desc, cnt)
    ex.last_status()
    for b, g in enumerate((g0, g1)):
        ko, do = oracle.orb_extract(g)
        assert _same(kps[b, :cnt[b]], desc[b, :cnt[b]], ko, do)


@pytest.mark.parametrize('seed', range(12))
def test_octree_kernel_random_candidates(ex, oracle, seed):
    """k_octree alone vs the oracle's list-based DistributeOctTree on random candidate sets,
    including heavy response ties and clustered points (phase-2 ordering, early-stop rule)."""
    rng = np.random.RandomState(100 + seed)
    level = seed % 8
    w, h = oracle.level_sizes(640, 480)[level]
    p = oracle.orb_params()
    N = int(p['per_level'][level])
    W, H = w - 32, h - 32                      # candidate coordinate range (relative to the border origin)
    n = int(rng.choice([1, 2, 5, N // 2, N, 3 * N, 8 * N, min(8000, 30 * N)]))
    # cell-interior coordinates start at 3 (a FAST candidate never sits in the 3-px apron)
    if seed % 3 == 0:      # clustered
        cx = rng.randint(3, W - 3, 6); cy = rng.randint(3, H - 3, 6)
        k = rng.randint(0, 6, n * 2)
        xs = np.clip(cx[k] + rng.randint(-25, 26, n * 2), 3, W - 4); ys = np.clip(cy[k] + rng.randint(-25, 26, n * 2), 3, H - 4)
    else:
        xs = rng.randint(3, W - 3, n * 2); ys = rng.randint(3, H - 3, n * 2)
    pts = np.unique(np.stack([ys, xs], 1), axis=0)[:n]          # distinct pixels
    rng.shuffle(pts)
    ys, xs = pts[:, 0], pts[:, 1]
    resp = rng.randint(7, 12 if seed % 2 else 200, len(xs))     # many ties when the range is narrow
    # oracle consumes candidates in the reference's cell-raster order
    lv = [l for l in range(8)][level]
    geomN = oracle.orb_params()
    # order by (cell row, cell col, y, x) as ComputeKeyPointsOctTree produces them
    width = float(w - 32); height = float(h - 32)
    ncols = int(width / 30); nrows = int(height / 30)
    wcell = int(np.ceil(width / ncols)); hcell = int(np.ceil(height / nrows))
    order = np.lexsort((xs, ys, (xs - 3) // wcell, (ys - 3) // hcell))
    xo, yo, ro = xs[order], ys[order], resp[order]
    sel = oracle.distribute_octree(xo.astype('f4'), yo.astype('f4'), ro.astype('f4'), 16, w - 16, 16, h - 16, N)
    exp = list(zip(xo[sel], yo[sel], ro[sel]))
    # kernel gets them shuffled (k_fast_cells appends unordered)
    perm = rng.permutation(len(xs))
    gx, gy, gs = ex.debug_run_octree(level, xs[perm], ys[perm], resp[perm])
    assert list(zip(gx, gy, gs)) == exp


def run_fused_pyramid_equals_per_level(lib, to_dev=lambda a: a):
    """k_pyramid (all levels in one launch, tiles chained through LDS) writes the same bytes as the per-level k_resize launches."""
    for (w, h, nl, sf) in ((640, 480, 8, 1.2), (752, 480, 8, 1.2), (512, 384, 6, 1.3), (644, 484, 4, 1.5)):
        rng = np.random.RandomState(w + h)
        imgs = rng.randint(0, 256, (2, h, w)).astype(np.uint8)
        e = ORBextractor(lib=lib, width=w, height=h, nlevels=nl, scaleFactor=sf, max_batch=2)
        levels = []
        from sg_slam_amd.capi import KP_DTYPE
        cap = e.capacity
        kps = to_dev(np.zeros((2, cap * 28), np.uint8)); desc = to_dev(np.zeros((2, cap, 32), np.uint8)); cnt = to_dev(np.zeros(2, 'i4')); dimg = to_dev(imgs)
        for unfused in (1, 0):
            lib.tap('sgx_orb_debug_set_unfused_pyramid')(unfused)
            e.extract_batch_dev(dimg, w, 2, kps, desc, cnt)
            e.last_status()
            levels.append([[e.debug_level(f, l) for l in range(1, nl)] for f in range(2)])
        e.close()
        for f in range(2):
            for a, b in zip(levels[0][f], levels[1][f]):
                assert a.shape == b.shape and (a == b).all(), (w, h)


def test_fused_pyramid_equals_per_level(emu):
    run_fused_pyramid_equals_per_level(emu)


def run_other_geometries(lib, oracle):
    """full extraction parity at image sizes other than 640x480 (level widths that are not multiples of the tile sizes, narrow last tiles)"""
    tex = synth.world_texture(7, 1024, 900)
    for (w, h) in ((752, 480), (516, 388), (320, 240)):
        img = np.ascontiguousarray(tex[100:100 + h, 50:50 + w])
        ko, do = oracle.orb_extract(img)
        e = ORBextractor(lib=lib, width=w, height=h)
        k, d = e(img)
        e.close()
        assert len(ko) > 500 and _same(k, d, ko, do), (w, h)


def test_other_geometries_emu(emu, oracle):
    run_other_geometries(emu, oracle)


def run_other_parameters(lib, oracle):
    """the reference's other shipped settings and a few off-grid ones: KITTI.yaml (1241x376 — a width that is not a multiple of 4 — 2000 features), a coarser / finer
    pyramid, other FAST thresholds; full extraction parity each"""
    tex = synth.world_texture(7, 1400, 1100)
    for (w, h, nf, sf, nl, ini, mn) in ((1241, 376, 2000, 1.2, 8, 20, 7), (800, 600, 3000, 1.3, 6, 30, 10), (640, 480, 1000, 1.1, 10, 12, 5)):
        img = np.ascontiguousarray(tex[40:40 + h, 60:60 + w])
        ko, do = oracle.orb_extract(img, nfeatures=nf, scale=sf, nlevels=nl, ini_th=ini, min_th=mn)
        e = ORBextractor(lib=lib, nfeatures=nf, scaleFactor=sf, nlevels=nl, iniThFAST=ini, minThFAST=mn, width=w, height=h)
        k, d = e(img)
        e.close()
        assert len(ko) > nf // 2 and _same(k, d, ko, do), (w, h, nf, sf, nl)


def test_other_parameters_emu(emu, oracle):
    run_other_parameters(emu, oracle)


def test_portrait_geometry_is_refused(emu, oracle):
    """a pyramid level more than twice as tall as wide gives DistributeOctTree nIni = round(w / h) = 0 root nodes; the reference then divides by zero and indexes an
    empty vector (ORBextractor.cc:544-566).  Product and oracle report the geometry instead of returning a level without keypoints / crashing."""
    with pytest.raises(Exception, match='unsupported'):
        ORBextractor(lib=emu, width=264, height=579)
    with pytest.raises(ValueError):
        oracle.orb_extract(np.zeros((579, 264), np.uint8))
    e = ORBextractor(lib=emu, width=300, height=560, nlevels=2)          # (300 - 32) / (560 - 32) = 0.51, (250 - 32) / (467 - 32) = 0.50 -> one root node per level: fine
    e.close()


def test_persistent_blur_walks_its_work_list(ex, oracle, stream_frames, monkeypatch):
    """k_blur_levels is persistent: with fewer workgroups than (tile, frame) work items every workgroup blurs several tiles, fetching the next tile's pixels while it
    works on the current one.  Eight workgroups for the ~600 tiles of two frames: descriptors still bit-exact."""
    monkeypatch.setenv('SGX_TUNE_ORB_BLUR_GRID', '8')
    g0, _, _ = stream_frames.frame(3); g1, _, _ = stream_frames.frame(11)
    from sg_slam_amd.capi import KP_DTYPE
    cap = ex.capacity
    kps = np.zeros((2, cap), KP_DTYPE); desc = np.zeros((2, cap, 32), np.uint8); cnt = np.zeros(2, 'i4')
    ex.extract_batch_dev(np.stack([g0, g1]), 640, 2, kps, desc, cnt)
    ex.last_status()
    for b, g in enumerate((g0, g1)):
        ko, do = oracle.orb_extract(g)
        assert _same(kps[b, :cnt[b]], desc[b, :cnt[b]], ko, do)
