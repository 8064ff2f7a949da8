"""ORBmatcher — host-side mirror of ORB_SLAM2::ORBmatcher (reference: src/sg-slam/include/ORBmatcher.h:41-89,
src/sg-slam/src/ORBmatcher.cc) over the C-ABI.  Implemented on the device so far:
  SearchByProjection(CurrentFrame, LastFrame, th, bMono)   ORBmatcher.cc:1332-1472
Frames are flattened dicts of numpy arrays:
  cur : keys[KP_DTYPE], desc[N,32] u8, uright[N] f4, Tcw[4,4] f4
  last: the same plus has_mp[N] u8, outlier[N] u8, xw[N,3] f4, obs[N] i4, mpdesc[N,32] u8
"""
import ctypes as C
import numpy as np
from .capi import Camera, _vp
from ._lib import load


def camera_struct(cam, width=640, height=480):
    return Camera(cam['fx'], cam['fy'], cam['cx'], cam['cy'], cam['bf'], cam.get('min_x', 0.0), cam.get('max_x', float(width)),
                  cam.get('min_y', 0.0), cam.get('max_y', float(height)))


class ORBmatcher:
    TH_HIGH = 100
    TH_LOW = 50
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, lib=None):
        self.lib = lib if lib is not None else load()
        self.mfNNratio = nnratio
        self.mbCheckOrientation = checkOri

    @staticmethod
    def DescriptorDistance(a, b):
        """ORBmatcher.cc:1649-1665 (host utility; the device kernels use the same popcount)."""
        return int(np.unpackbits(np.bitwise_xor(np.asarray(a, np.uint8), np.asarray(b, np.uint8))).sum())

    def SearchByProjection(self, cur, last, th, bMono, cam, scale_factors):
        ck = np.ascontiguousarray(cur['keys']); cd = np.ascontiguousarray(cur['desc'], np.uint8); cu = np.ascontiguousarray(cur['uright'], 'f4')
        cT = np.ascontiguousarray(cur['Tcw'], 'f4').reshape(16)
        lk = np.ascontiguousarray(last['keys']); lh = np.ascontiguousarray(last['has_mp'], np.uint8); lo = np.ascontiguousarray(last['outlier'], np.uint8)
        lx = np.ascontiguousarray(last['xw'], 'f4'); lb = np.ascontiguousarray(last['obs'], 'i4'); lm = np.ascontiguousarray(last['mpdesc'], np.uint8)
        lT = np.ascontiguousarray(last['Tcw'], 'f4').reshape(16)
        sf = np.ascontiguousarray(scale_factors, 'f4')
        match = np.full(max(len(ck), 1), -1, 'i4'); n = np.zeros(1, 'i4')
        cs = camera_struct(cam)
        rc = self.lib.dll.sgx_match_project_frame(len(ck), _vp(ck), _vp(cd), _vp(cu), _vp(cT), len(lk), _vp(lk), _vp(lh), _vp(lo), _vp(lx),
                                                  _vp(lb), _vp(lm), _vp(lT), C.byref(cs), _vp(sf), len(sf), float(th), int(bMono),
                                                  int(self.mbCheckOrientation), _vp(match), _vp(n))
        self.lib.check(rc, 'sgx_match_project_frame')
        cur['match'] = match[:len(ck)]
        return int(n[0])
