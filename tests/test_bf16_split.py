"""CPU: the host half of the bf16x3 scheme (sg_slam_amd/csrc/sgx_det_bf16.h) — every fp32 weight is split EXACTLY into three bf16 terms (round to nearest even, residuals exact)
and laid out as the A operand of v_mfma_f32_32x32x16_bf16, [k16 step][term][half][oc (ld)][8].  The device half (k_conv_pw3 / the split of the activations in registers) is checked on
the GPU against the oracle and against the exact-fp32 plan (tests/test_detector_gpu.py)."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def probe(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('bf16') / 'libbf16probe.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-DSGX_EMU', '-Wno-unused-function', '-Wno-unused-variable', '-Wno-unknown-pragmas',
                           os.path.join(ROOT, 'tests', 'host', 'bf16_split_probe.cpp'), '-o', so])
    return C.CDLL(so)


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def test_bf16_round_to_nearest_even(probe):
    probe.probe_bf16_rne.restype = C.c_ushort; probe.probe_bf16_rne.argtypes = [C.c_float]
    rng = np.random.RandomState(0)
    xs = np.concatenate([rng.randn(2000).astype('f4') * 10.0 ** rng.randint(-20, 20, 2000), np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.3895314e38, 1e-40], 'f4')])
    for x in xs:
        h = probe.probe_bf16_rne(float(x)); u = np.float32(x).view(np.uint32)
        lo, hi = np.uint32(u & 0xFFFF0000), np.uint32((u & 0xFFFF0000) + 0x10000)
        dl, dh = abs(float(x) - float(lo.view(np.float32))), abs(float(hi.view(np.float32)) - float(x))
        want = lo if dl < dh else hi if dh < dl else (lo if ((lo >> 16) & 1) == 0 else hi)        # ties to even
        assert h == int(want >> 16), (x, h, want)


@pytest.mark.parametrize('outc,K', [(40, 120), (112, 80), (21, 7), (160, 672), (33, 16)])
def test_split_is_exact_and_laid_out_as_the_mfma_operand(probe, outc, K):
    rng = np.random.RandomState(outc * 1000 + K)
    w = (rng.randn(outc, K) * 10.0 ** rng.randint(-6, 4, (outc, K))).astype('f4')
    w[0, 0] = 0.0; w[-1, -1] = np.float32(1.0) + np.float32(2.0) ** -23
    ldw = ((outc + 31) // 32) * 32 + 128
    nks = (K + 15) // 16
    dst = np.full(nks * 6 * ldw * 8, 0xFFFF, np.uint16)
    assert probe.probe_split_weights(w.ctypes.data_as(C.c_void_p), outc, K, ldw, dst.ctypes.data_as(C.c_void_p)) == nks
    d = dst.reshape(nks, 3, 2, ldw, 8)
    terms = np.zeros((3, outc, K), np.float32)
    for k in range(K):
        terms[:, :, k] = bf16_to_f32(d[k >> 4, :, (k >> 3) & 1, :outc, k & 7])
    t0, t1, t2 = terms.astype(np.float64)
    assert (t0 + t1 + t2 == w.astype(np.float64)).all()                                   # the three terms ARE the weight
    assert (np.abs(t1) <= np.abs(w.astype(np.float64)) * 2.0 ** -8 + 1e-300).all() and (np.abs(t2) <= np.abs(w.astype(np.float64)) * 2.0 ** -16 + 1e-300).all()
    # everything outside (oc >= outc, k >= K) is zero: padded rows / columns meet clamped activations on the device
    mask = np.ones(d.shape, bool)
    for k in range(K): mask[k >> 4, :, (k >> 3) & 1, :outc, k & 7] = False
    assert (d[mask] == 0).all()
