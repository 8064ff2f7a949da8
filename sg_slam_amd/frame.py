"""Per-frame glue between the accelerated stages, mirroring the reference's Frame helpers
(src/sg-slam/src/Frame.cc:893-932) over the C-ABI batch entry points.  Pointers may be torch
CUDA tensors (product) or numpy arrays (kernel-logic emulator in tests)."""
import ctypes as C
from .capi import _vp
from .matcher import camera_struct


def stereo_from_rgbd_batch(lib, batch, cap, d_keys, d_n, d_depth_u16, width, height, depth_map_factor, bf, d_uright, d_zdepth, stream=None):
    lib.check(lib.dll.sgx_frame_stereo_from_rgbd_batch_dev(batch, cap, _vp(d_keys), _vp(d_n), _vp(d_depth_u16), width, height,
                                                           float(depth_map_factor), float(bf), _vp(d_uright), _vp(d_zdepth), _vp(stream)),
              'sgx_frame_stereo_from_rgbd_batch_dev')


def unproject_batch(lib, batch, cap, d_keys, d_n, d_zdepth, d_Tcw, cam, d_xw, d_has, stream=None):
    cs = camera_struct(cam)
    lib.check(lib.dll.sgx_frame_unproject_batch_dev(batch, cap, _vp(d_keys), _vp(d_n), _vp(d_zdepth), _vp(d_Tcw), C.byref(cs), _vp(d_xw),
                                                    _vp(d_has), _vp(stream)), 'sgx_frame_unproject_batch_dev')
