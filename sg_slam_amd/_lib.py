"""Product loader: ONLY sg_slam_amd/libsgx.so (hipcc, gfx950).  No CPU fallback — a missing
library is a hard error (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C sg_slam_amd/csrc`)."""
import os
from .capi import SgxLib

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path():
    return os.path.join(_HERE, 'libsgx.so')


def load():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError(f'sg_slam_amd: HIP library {p} is missing; the product has no CPU fallback. '
                               f'Run `make -C sg_slam_amd/csrc` (hipcc --offload-arch=gfx950).')
        _LIB = SgxLib(p)
    return _LIB
