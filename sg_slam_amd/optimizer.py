"""Optimizer — host-side mirror of ORB_SLAM2::Optimizer (reference: src/sg-slam/include/Optimizer.h:40-58)
over the C-ABI.  Implemented on the device so far:
  PoseOptimization(pFrame)     src/sg-slam/src/Optimizer.cc:239-451
The Frame is a flattened dict: keys (mvKeysUn, KP_DTYPE), uright, has_mp, xw (map-point world
positions per keypoint), Tcw (4x4 f32).  PoseOptimization updates frame['Tcw'] and frame['outlier']
in place and returns the inlier count, like the reference mutates pFrame."""
import ctypes as C
import numpy as np
from .capi import _vp
from .matcher import camera_struct
from ._lib import load


class Optimizer:
    @staticmethod
    def PoseOptimization(frame, cam, inv_level_sigma2, lib=None):
        lib = lib if lib is not None else load()
        k = np.ascontiguousarray(frame['keys']); ur = np.ascontiguousarray(frame['uright'], 'f4')
        has = np.ascontiguousarray(frame['has_mp'], np.uint8); xw = np.ascontiguousarray(frame['xw'], 'f4')
        T = np.ascontiguousarray(frame['Tcw'], 'f4').reshape(16).copy()
        is2 = np.ascontiguousarray(inv_level_sigma2, 'f4')
        n = len(k)
        out = np.zeros(max(n, 1), np.uint8); ninl = np.zeros(1, 'i4')
        cs = camera_struct(cam)
        lib.check(lib.dll.sgx_pose_optimization(n, _vp(k), _vp(ur), _vp(has), _vp(xw), _vp(is2), len(is2), C.byref(cs), _vp(T), _vp(out), _vp(ninl)),
                  'sgx_pose_optimization')
        frame['Tcw'] = T.reshape(4, 4)
        frame['outlier'] = out[:n]
        return int(ninl[0])
