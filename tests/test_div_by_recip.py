"""sgx_div_by_recip (quotient from a shared reciprocal, used by the pose / BA kernels for p/z and q/|q|) must equal the IEEE
division the reference performs, bit for bit: the device source compiled for the host against `a / b`."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#define SGX_EMU
#include "%s/sg_slam_amd/csrc/sgx_se3.h"
thread_local sgx_dim3 blockIdx, blockDim, gridDim;
extern "C" long check(long n, const double *a, const double *b) {
  long bad = 0;
  for (long i = 0; i < n; i++) { const double r = 1.0 / b[i]; const double q = sgx_div_by_recip(a[i], b[i], r), t = a[i] / b[i];
    if (memcmp(&q, &t, 8) != 0 && !(q != q && t != t)) bad++; }
  return bad; }
''' % ROOT


def test_div_by_recip_is_ieee_division(tmp_path):
    src = tmp_path / 't.cpp'; so = tmp_path / 't.so'
    src.write_text(SRC)
    subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-std=c++17', '-shared', '-fPIC', '-Wno-unknown-pragmas', str(src), '-o', str(so), '-lm'])
    lib = C.CDLL(str(so)); lib.check.restype = C.c_long
    rng = np.random.RandomState(5)
    n = 2_000_000
    # the kernels' ranges (metres, pixels, unit quaternions) ...
    a = rng.uniform(-50, 50, n); b = rng.uniform(0.05, 60, n) * rng.choice([-1.0, 1.0], n)
    assert lib.check(C.c_long(n), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
    # ... random mantissas over 40 binades either side ...
    a = np.ldexp(rng.uniform(1, 2, n), rng.randint(-40, 40, n)) * rng.choice([-1.0, 1.0], n)
    b = np.ldexp(rng.uniform(1, 2, n), rng.randint(-40, 40, n)) * rng.choice([-1.0, 1.0], n)
    assert lib.check(C.c_long(n), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
    # ... and the special operands (a zero / infinite divisor must give the IEEE +-inf / 0 / NaN, not a NaN from the correction step)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 3.5, 1e300, 1e-300])
    a, b = [x.ravel().copy() for x in np.meshgrid(sp, sp)]
    assert lib.check(C.c_long(len(a)), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == 0
