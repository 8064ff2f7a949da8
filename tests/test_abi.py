"""The C-ABI library loads and exports every symbol include/sgx.h declares (no compute calls)."""
import ctypes
import os
import re
import pytest
from sg_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header='sgx.h'):
    src = open(os.path.join(ROOT, 'include', header)).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sgx_[a-z0-9_]+)\s*\(', src)))


def test_binding_covers_header():
    assert sorted(capi.SYMBOLS) == _declared()
    assert sorted(capi.TAP_SYMBOLS) == _declared('sgx_debug.h')
    assert not [s for s in capi.SYMBOLS if 'debug' in s]


def test_product_library_exports_all_symbols():
    so = os.path.join(ROOT, 'sg_slam_amd', 'libsgx.so')
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    dll = ctypes.CDLL(so)           # loads without a GPU (no HIP call happens at load time)
    for s in _declared():
        assert hasattr(dll, s), s
    dll.sgx_version.restype = ctypes.c_char_p
    assert b'gfx950' in dll.sgx_version()


def test_product_library_has_no_taps_and_reads_no_environment():
    """the test / tuning taps (include/sgx_debug.h) and the SGX_* environment switches are compiled out of the product build"""
    import subprocess
    so = os.path.join(ROOT, 'sg_slam_amd', 'libsgx.so')
    dll = ctypes.CDLL(so)
    for s in _declared('sgx_debug.h'):
        assert not hasattr(dll, s), s
    exported = subprocess.check_output(['nm', '-D', '--defined-only', so]).decode()
    assert 'debug' not in exported.lower()
    assert 'getenv' not in subprocess.check_output(['nm', '-D', '--undefined-only', so]).decode()
    text = open(so, 'rb').read()
    assert not re.findall(rb'SGX_(?:TUNE|DET|IRB|BA|PW|DW|LK|TRK|FB|ENV)[A-Z0-9_]*\x00', text)      # no switch name survives as a string


def test_collective_entry_needs_no_rccl_at_load_time():
    """sgx_dist_* (BASELINE config 5 from the C++ host) load RCCL on first use: the library itself must not depend on it (single-GPU programs, this CPU container), and
    the emulator reports the entry as unsupported instead of pretending"""
    import subprocess
    so = os.path.join(ROOT, 'sg_slam_amd', 'libsgx.so')
    needed = subprocess.check_output(['readelf', '-d', so]).decode()
    assert 'rccl' not in needed.lower() and 'nccl' not in needed.lower()
    emu = os.path.join(ROOT, 'tests', 'emu', 'libsgx_emu.so')
    if os.path.exists(emu):
        dll = ctypes.CDLL(emu)
        buf = ctypes.create_string_buffer(128)
        assert dll.sgx_dist_unique_id(buf) != 0
        h = ctypes.c_void_p()
        assert dll.sgx_dist_create(buf, 1, 0, ctypes.byref(h)) != 0 and not h.value


def test_tap_builds_export_the_taps():
    for so in (os.path.join(ROOT, 'tests', 'taps', 'libsgx_taps.so'), os.path.join(ROOT, 'tests', 'emu', 'libsgx_emu.so')):
        if not os.path.exists(so):
            pytest.skip(so + ' not built')
        dll = ctypes.CDLL(so)
        for s in _declared() + _declared('sgx_debug.h'):
            assert hasattr(dll, s), (so, s)


def test_loader_has_no_fallback(monkeypatch, tmp_path):
    from sg_slam_amd import _lib
    monkeypatch.setattr(_lib, '_LIB', None)
    monkeypatch.setattr(_lib, '_HERE', str(tmp_path))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()
