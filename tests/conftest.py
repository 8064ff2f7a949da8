import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAVE_GPU = _have_gpu()


@pytest.fixture(scope='session', autouse=True)
def _lds_polluter():
    """SGX_TEST_POLLUTE=<KB> (GPU box, optional): a side stream keeps overwriting the LDS of every CU with changing garbage while the tests run (tools/lds_pollute).  LDS is not
    cleared between kernels, so a kernel that reads LDS it never wrote only shows as a difference when the leftovers change; round 6 found one that way.
    SGX_TEST_CORUN=<kind bits> (GPU box, optional): the side stream runs tools/lds_pollute's k_corun instead (2 = dense bf16 matrix products on every CU): round 6 found that
    compiler-generated packed fp32 instructions of a co-resident wave go wrong beside them (profiles/r6_lk_priority_diagnosis.md); the whole tier must pass with it on."""
    kb = os.environ.get('SGX_TEST_POLLUTE'); kind = os.environ.get('SGX_TEST_CORUN')
    if not (kb or kind) or not HAVE_GPU:
        yield; return
    import ctypes, threading
    so = os.path.join(ROOT, 'tools', 'lds_pollute', 'liblds_pollute.so')
    if not os.path.exists(so):
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-fPIC', '-shared', so.replace('liblds_pollute.so', 'lds_pollute.hip'), '-o', so])
    lib = ctypes.CDLL(so); stop = threading.Event()
    def run():
        import torch
        torch.cuda.set_device(0); i = 0
        while not stop.is_set():
            if kind: lib.corun_launch(1024, 2000, int(kind), 8, 3); lib.corun_sync()      # k_corun: SGX_TEST_CORUN=2 is a dense stream of bf16 matrix products on every CU
            else: lib.lds_pollute(ctypes.c_uint32(0x7fc00000 + 7919 * i), int(kb), 64, 1024); lib.lds_pollute_sync()
            i += 1
    th = threading.Thread(target=run, daemon=True); th.start()
    yield
    stop.set(); th.join(timeout=10)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope='session')
def emu():
    """Kernel-logic emulator: the product kernels compiled with -DSGX_EMU for the host (tests only)."""
    from sg_slam_amd.capi import SgxLib
    so = os.path.join(ROOT, 'tests', 'emu', 'libsgx_emu.so')
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'sg_slam_amd', 'csrc'), 'emu'])
    lib = SgxLib(so)
    assert 'EMULATOR' in lib.version()
    return lib


@pytest.fixture(scope='session')
def gpulib():
    """The product library on a real GPU; must be the HIP build."""
    if not HAVE_GPU:
        pytest.skip('no GPU')
    import sg_slam_amd
    lib = sg_slam_amd.load()
    assert 'gfx950' in lib.version() and not lib.has_taps
    return lib


@pytest.fixture(scope='session')
def gpulib_taps():
    """The product sources built with -DSGX_DEBUG_TAPS (tests/taps/libsgx_taps.so): the same kernels plus the entries of include/sgx_debug.h (blob / pyramid read-backs,
    plan selection) and the SGX_* environment switches, which the product library does not have.  For the GPU tests that look INSIDE a stage; tests only."""
    if not HAVE_GPU:
        pytest.skip('no GPU')
    from sg_slam_amd.capi import SgxLib
    so = os.path.join(ROOT, 'tests', 'taps', 'libsgx_taps.so')
    if not os.path.exists(so):          # normally built by __graft_entry__.build() and shipped with the tree; a box that has hipcc can build it (about two minutes)
        subprocess.call(['make', '-s', '-j16', '-C', os.path.join(ROOT, 'sg_slam_amd', 'csrc'), 'taps'])
    if not os.path.exists(so):
        pytest.fail('tests/taps/libsgx_taps.so not built (make -C sg_slam_amd/csrc taps)')
    lib = SgxLib(so)
    assert 'gfx950' in lib.version() and lib.has_taps
    return lib


@pytest.fixture(scope='session')
def gpulib_nofma():
    """The product sources built with -DSGX_FP_CONTRACT_OFF (no multiply-add fusion in the fp64 solvers: the reference's rounding); tests only."""
    if not HAVE_GPU:
        pytest.skip('no GPU')
    from sg_slam_amd.capi import SgxLib
    so = os.path.join(ROOT, 'tests', 'nofma', 'libsgx_nofma.so')
    if not os.path.exists(so):
        pytest.skip('tests/nofma/libsgx_nofma.so not built (make -C sg_slam_amd/csrc nofma)')
    lib = SgxLib(so)
    assert 'gfx950' in lib.version()
    return lib


@pytest.fixture(scope='session')
def stream_frames():
    from sg_slam_amd import synth
    S = synth.PlaneStream(seed=1234)
    return S
