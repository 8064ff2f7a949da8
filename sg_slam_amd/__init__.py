"""sg_slam_amd — MI355X-native (gfx950) tracking hot path for SG-SLAM.

Host-side mirror of the reference operator interfaces over the C-ABI in include/sgx.h:
  ORBextractor  (reference: src/sg-slam/include/ORBextractor.h)
The HIP library (sg_slam_amd/libsgx.so) is mandatory: importing the operators without it raises.
"""
from .capi import SgxLib, SgxError, KP_DTYPE  # noqa: F401
from ._lib import load  # noqa: F401
