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

    def SearchByProjectionLocal(self, F, local_map, th, cam, scale_factors, viewing_cos_limit=0.5):
        """Tracking::SearchLocalPoints' inner work: Frame::isInFrustum(pMP, 0.5) for every local map point followed by
        ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (ORBmatcher.cc:45-129) with this matcher's
        nnratio.  F: keys, desc, uright, Tcw [, mp_obs]; local_map: xw, normal, min_dist, max_dist, desc, obs, skip.
        Sets F['match_local'] (index into local_map or -1) and local_map['in_view']; returns nmatches."""
        ck = np.ascontiguousarray(F['keys']); cd = np.ascontiguousarray(F['desc'], np.uint8); cu = np.ascontiguousarray(F['uright'], 'f4')
        cT = np.ascontiguousarray(F['Tcw'], 'f4').reshape(16); co = np.ascontiguousarray(F.get('mp_obs', np.full(len(ck), -1)), 'i4')
        lm = local_map
        xw = np.ascontiguousarray(lm['xw'], 'f4'); nr = np.ascontiguousarray(lm['normal'], 'f4')
        mnd = np.ascontiguousarray(lm['min_dist'], 'f4'); mxd = np.ascontiguousarray(lm['max_dist'], 'f4')
        md = np.ascontiguousarray(lm['desc'], np.uint8); mo = np.ascontiguousarray(lm['obs'], 'i4'); ms = np.ascontiguousarray(lm['skip'], np.uint8)
        sf = np.ascontiguousarray(scale_factors, 'f4')
        nc, nm = len(ck), len(xw)
        if nc == 0 or nm == 0:              # nothing to project / nothing to match against: the reference's loops do not execute (ORBmatcher.cc:52-127)
            F['match_local'] = np.full(nc, -1, 'i4'); local_map['in_view'] = np.zeros(nm, np.uint8)
            return 0
        match = np.full(nc, -1, 'i4'); n = np.zeros(1, 'i4'); inview = np.zeros(nm, np.uint8)
        cs = camera_struct(cam)
        rc = self.lib.dll.sgx_match_project_local(nc, _vp(ck), _vp(cd), _vp(cu), _vp(cT), _vp(co), nm, _vp(xw), _vp(nr), _vp(mnd), _vp(mxd), _vp(md), _vp(mo), _vp(ms),
                                                  C.byref(cs), _vp(sf), len(sf), float(np.log(np.float32(sf[1]))), float(th), float(self.mfNNratio),
                                                  float(viewing_cos_limit), _vp(match), _vp(n), _vp(inview))
        self.lib.check(rc, 'sgx_match_project_local')
        F['match_local'] = match; local_map['in_view'] = inview
        return int(n[0])
