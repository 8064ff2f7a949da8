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
