"""The C-ABI library loads and exports every symbol include/sgx.h declares (no compute calls)."""
import ctypes
import os
import re
import pytest
from sg_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'sgx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sgx_[a-z0-9_]+)\s*\(', src)))


def test_binding_covers_header():
    assert sorted(capi.SYMBOLS) == _declared()


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


def test_loader_has_no_fallback(monkeypatch, tmp_path):
    from sg_slam_amd import _lib
    monkeypatch.setattr(_lib, '_LIB', None)
    monkeypatch.setattr(_lib, '_HERE', str(tmp_path))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()
