"""Mask inputs (tier N1): kernel LOGIC of the LK tracker and the RANSAC fundamental matrix against the oracle, through the C-ABI of the
kernel-logic emulator (CPU tier; tests/test_flow_gpu.py repeats the same assertions on the device)."""
import numpy as np
import pytest
import flow_cases as fc


def test_oracle_pyrdown_kat(oracle):
    """cv::pyrDown known answers: a constant image stays constant; an impulse gives the outer product [1 4 6 4 1]^2 / 256 (rounded)"""
    assert (oracle.pyr_down(np.full((40, 60), 77, np.uint8)) == 77).all()
    img = np.zeros((41, 41), np.uint8); img[20, 20] = 255
    d = oracle.pyr_down(img)
    k = np.array([1, 4, 6, 4, 1])
    exp = (np.outer(k, k)[::2, ::2] * 255 + 128) >> 8          # taps that land on even source coordinates
    assert (d[9:12, 9:12] == exp).all() and d.sum() == exp.sum()


def test_oracle_scharr_kat(oracle):
    """a horizontal ramp: dx = 32 * slope inside (3+10+3 rows x 2 columns), 0 at the reflected borders; dy = 0"""
    img = np.tile((np.arange(50) * 3).astype(np.uint8), (30, 1))
    d = oracle.scharr_deriv(img)
    assert (d[:, 1:-1, 0] == 16 * 6).all() and (d[:, 0, 0] == 0).all() and (d[:, -1, 0] == 0).all() and (d[..., 1] == 0).all()


def test_oracle_rng_kat(oracle):
    """cv::RNG is a multiply-with-carry generator: state' = (u32)state * 4164903690 + (state >> 32)"""
    import ctypes as C
    out = np.zeros(5, 'i4')
    oracle.lib().orc_rng_sequence.restype = C.c_uint
    oracle.lib().orc_rng_sequence(C.c_uint64(0xffffffffffffffff), C.c_int(5), C.c_int(1000), out.ctypes.data_as(C.c_void_p))
    s = 0xffffffffffffffff; exp = []
    for _ in range(5):
        s = ((s & 0xffffffff) * 4164903690 + (s >> 32)) & 0xffffffffffffffff
        exp.append((s & 0xffffffff) % 1000)
    assert list(out) == exp


def test_oracle_cubic_kat(oracle):
    n, r = oracle.solve_cubic([1, -6, 11, -6])
    assert n == 3 and np.allclose(sorted(r), [1, 2, 3], atol=1e-12)
    n, r = oracle.solve_cubic([1, 0, 1, 0])
    assert n == 1 and abs(r[0]) < 1e-12
    n, r = oracle.solve_cubic([0, 1, -3, 2])
    assert n == 2 and np.allclose(sorted(r[:2]), [1, 2])


def test_oracle_7point_kat(oracle):
    """exact correspondences of a known geometry: one of the roots is the generating F; every root is singular and satisfies the 7 constraints;
    the solution set equals the one obtained from numpy's SVD null space (basis independence)"""
    x1, x2 = fc.two_view(7, 5, 0, noise=0.0)
    Fs = oracle.fm_run7point(x1, x2)
    assert 1 <= len(Fs) <= 3
    A = np.array([[b[0] * a[0], b[0] * a[1], b[0], b[1] * a[0], b[1] * a[1], b[1], a[0], a[1], 1] for a, b in zip(x1.astype('f8'), x2.astype('f8'))])
    _, _, Vt = np.linalg.svd(A)
    f1, f2 = Vt[7], Vt[8]
    ls = np.array([-1.0, 0.0, 1.0, 2.0])
    co = np.polyfit(ls, [np.linalg.det((l * f1 + (1 - l) * f2).reshape(3, 3)) for l in ls], 3)
    sol = [(r.real * f1 + (1 - r.real) * f2).reshape(3, 3) for r in np.roots(co) if abs(r.imag) < 1e-9]
    sol = [s / s[2, 2] for s in sol]
    for F in Fs:
        assert abs(np.linalg.det(F / np.abs(F).max())) < 1e-9
        assert np.abs(A @ F.reshape(9)).max() < 1e-6 * np.abs(A).max() * np.abs(F).max()
        assert min(np.abs(F - s).max() / np.abs(s).max() for s in sol) < 1e-8


def test_oracle_ransac_rejects_outliers(oracle):
    x1, x2 = fc.two_view(400, 1, 4)
    ok, F, mask, st = oracle.find_fundamental_ransac(x1, x2)
    assert ok == 1 and mask[::4].mean() < 0.1 and np.delete(mask, np.s_[::4]).mean() > 0.9 and 1 <= st[0] <= 1000


def test_pyramid_emu(emu, oracle):
    fc.check_pyramid(emu, oracle)


def test_lk_pair_emu(emu, oracle):
    fc.check_lk_pair(emu, oracle)


def test_lk_textureless_emu(emu, oracle):
    fc.check_lk_textureless(emu, oracle)


def test_lk_stream_emu(emu, oracle):
    fc.check_lk_stream(emu, oracle, lambda a: a)


def test_ransac_host_emu(emu, oracle):
    fc.check_ransac_host(emu, oracle)


def test_ransac_batch_emu(emu, oracle):
    fc.check_ransac_batch(emu, oracle, lambda a: a)
