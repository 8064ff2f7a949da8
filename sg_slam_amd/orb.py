"""ORBextractor — host-side mirror of ORB_SLAM2::ORBextractor
(reference: src/sg-slam/include/ORBextractor.h:45-111, src/sg-slam/src/ORBextractor.cc:411-471, :1045-1106)
over the C-ABI (include/sgx.h).  Same constructor arguments, same getters, `__call__` == operator()."""
import ctypes as C
import numpy as np
from .capi import OrbConfig, KP_DTYPE, _vp
from ._lib import load


class ORBextractor:
    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 width=640, height=480, max_batch=1, lib=None):
        self.lib = lib if lib is not None else load()
        self.cfg = OrbConfig(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, max_batch)
        h = C.c_void_p()
        self.lib.check(self.lib.dll.sgx_orb_create(C.byref(self.cfg), C.byref(h)), 'sgx_orb_create')
        self.h = h
        self.nlevels, self.nfeatures, self.width, self.height, self.max_batch = nlevels, nfeatures, width, height, max_batch
        self.capacity = self.lib.dll.sgx_orb_keypoint_capacity(self.h)
        t = [np.zeros(nlevels, 'f4') for _ in range(4)] + [np.zeros(nlevels, 'i4')]
        self.lib.check(self.lib.dll.sgx_orb_get_tables(self.h, *[_vp(a) for a in t]), 'sgx_orb_get_tables')
        self.mvScaleFactor, self.mvInvScaleFactor, self.mvLevelSigma2, self.mvInvLevelSigma2, self.mnFeaturesPerLevel = t

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dll.sgx_orb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # reference getters (ORBextractor.h:63-85)
    def GetLevels(self): return self.nlevels
    def GetScaleFactor(self): return float(self.cfg.scale_factor)
    def GetScaleFactors(self): return self.mvScaleFactor
    def GetInverseScaleFactors(self): return self.mvInvScaleFactor
    def GetScaleSigmaSquares(self): return self.mvLevelSigma2
    def GetInverseScaleSigmaSquares(self): return self.mvInvLevelSigma2
    def GetnFeatures(self): return self.nfeatures

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors): host gray image -> (keypoints[KP_DTYPE], desc[N,32])."""
        img = np.ascontiguousarray(image, np.uint8)
        if img.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)      # ORBextractor.cc:1048
        assert img.shape == (self.height, self.width)
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int(0)
        self.lib.check(self.lib.dll.sgx_orb_extract(self.h, _vp(img), self.width, _vp(kps), _vp(desc), self.capacity, C.byref(n)),
                       'sgx_orb_extract')
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch_dev(self, d_gray, pitch, batch, d_kps, d_desc, d_count, stream=None):
        """Batched device-resident call (pointers: ints, torch tensors, or numpy arrays under the emulator)."""
        self.lib.check(self.lib.dll.sgx_orb_extract_batch_dev(self.h, _vp(d_gray), pitch, batch, _vp(d_kps), _vp(d_desc),
                                                              _vp(d_count), self.capacity, _vp(stream)), 'sgx_orb_extract_batch_dev')

    def last_status(self, stream=None):
        self.lib.check(self.lib.dll.sgx_orb_last_status(self.h, _vp(stream)), 'sgx_orb_last_status')

    # diagnostic taps
    def debug_level(self, frame, level):
        w, h, s = C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.check(self.lib.tap('sgx_orb_debug_level_geometry')(self.h, level, C.byref(w), C.byref(h), C.byref(s)))
        out = np.zeros((h.value, w.value), np.uint8)
        self.lib.check(self.lib.tap('sgx_orb_debug_read_level')(self.h, frame, level, _vp(out)), 'debug_read_level')
        return out

    def debug_candidates(self, frame, level, cap=8192):
        x = np.zeros(cap, 'i4'); y = np.zeros(cap, 'i4'); s = np.zeros(cap, 'i4'); n = C.c_int(0)
        self.lib.check(self.lib.tap('sgx_orb_debug_read_candidates')(self.h, frame, level, _vp(x), _vp(y), _vp(s), cap, C.byref(n)))
        return x[:n.value].copy(), y[:n.value].copy(), s[:n.value].copy()

    def debug_run_octree(self, level, x, y, score):
        packed = (np.asarray(x, np.uint32) | (np.asarray(y, np.uint32) << 12) | (np.asarray(score, np.uint32) << 24)).astype(np.uint32)
        packed = np.ascontiguousarray(packed)
        out = np.zeros(2048, np.uint32); n = C.c_int(0)
        self.lib.check(self.lib.tap('sgx_orb_debug_run_octree')(self.h, level, _vp(packed), len(packed), _vp(out), 2048, C.byref(n)), 'debug_run_octree')
        o = out[:n.value]
        return (o & 0xFFF).astype('i4'), ((o >> 12) & 0xFFF).astype('i4'), (o >> 24).astype('i4')
