"""Mask inputs — host-side mirror of what Frame::RmDynamicPointWithSemanticAndGeometry computes before its erase loop
(reference: src/sg-slam/src/Frame.cc:430-472) over the C-ABI (include/sgx.h):

  OpticalFlowLK   cv::calcOpticalFlowPyrLK(imGray, imGrayPre, Curpoint, Prepoint, State, Err, Size(21,21), 3, TermCriteria(ITER|EPS, 30, 0.01))
  find_fundamental_mat / fundamental_ransac_batch_dev   cv::findFundamentalMat(cur, prev, FM_RANSAC, 1.0, 0.99) + the :454-467 pair selection
"""
import ctypes as C
import numpy as np
from .capi import FlowConfig, _vp
from ._lib import load


class OpticalFlowLK:
    """Owns the current / previous image pyramids like the file-scope `imGrayPre` (Frame.cc:31, :155-163)."""

    def __init__(self, width=640, height=480, max_batch=1, win_size=21, max_level=3, max_count=30, epsilon=0.01, lib=None):
        self.lib = lib if lib is not None else load()
        self.cfg = FlowConfig(width, height, max_batch, win_size, max_level, max_count, epsilon)
        h = C.c_void_p()
        self.lib.check(self.lib.dll.sgx_flow_create(C.byref(self.cfg), C.byref(h)), 'sgx_flow_create')
        self.h = h
        self.width, self.height, self.max_batch = width, height, max_batch
        self.levels = self.lib.dll.sgx_flow_levels(self.h)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dll.sgx_flow_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.lib.check(self.lib.dll.sgx_flow_reset(self.h), 'sgx_flow_reset')

    def __call__(self, gray_from, gray_to, pts):
        """calcOpticalFlowPyrLK(gray_from, gray_to, pts) on host images: (next_pts[n,2] f32, status[n] u8)."""
        a = np.ascontiguousarray(gray_from, np.uint8); b = np.ascontiguousarray(gray_to, np.uint8)
        assert a.shape == b.shape == (self.height, self.width)
        p = np.ascontiguousarray(pts, 'f4').reshape(-1, 2); n = len(p)
        out = np.zeros((max(n, 1), 2), 'f4'); st = np.zeros(max(n, 1), np.uint8)
        self.lib.check(self.lib.dll.sgx_flow_lk(self.h, _vp(a), _vp(b), self.width, _vp(p), n, _vp(out), _vp(st)), 'sgx_flow_lk')
        return out[:n], st[:n]

    def lk_batch_dev(self, d_gray, pitch, batch, d_keys, d_n, cap, d_prev_xy, d_status=None, stream=None):
        """Streaming form: pyramid of the new frames, flow into the previous call's frames, swap.  Returns True when a previous batch existed."""
        have = C.c_int32(0)
        self.lib.check(self.lib.dll.sgx_flow_lk_batch_dev(self.h, _vp(d_gray), pitch, batch, _vp(d_keys), _vp(d_n), cap, _vp(d_prev_xy), _vp(d_status),
                                                          C.byref(have), _vp(stream)), 'sgx_flow_lk_batch_dev')
        return bool(have.value)

    def debug_level(self, slot, frame, level):
        w, h = C.c_int32(), C.c_int32()
        self.lib.check(self.lib.tap('sgx_flow_debug_level_size')(self.h, level, C.byref(w), C.byref(h)))
        img = np.zeros((h.value, w.value), np.uint8)
        self.lib.check(self.lib.tap('sgx_flow_debug_read_level')(self.h, slot, frame, level, _vp(img)), 'flow debug_read_level')
        return img


def find_fundamental_mat(pts1, pts2, threshold=1.0, confidence=0.99, lib=None):
    """cv::findFundamentalMat(pts1, pts2, FM_RANSAC, threshold, confidence): (ok, F[3,3] f64, stats[4])."""
    lib = lib if lib is not None else load()
    a = np.ascontiguousarray(pts1, 'f4').reshape(-1, 2); b = np.ascontiguousarray(pts2, 'f4').reshape(-1, 2)
    assert len(a) == len(b)
    F = np.zeros(9, 'f8'); ok = np.zeros(1, 'i4'); stats = np.zeros(4, 'i4')
    lib.check(lib.dll.sgx_find_fundamental_mat(_vp(a), _vp(b), len(a), threshold, confidence, _vp(F), _vp(ok), _vp(stats)), 'sgx_find_fundamental_mat')
    return int(ok[0]), F.reshape(3, 3), stats


def fundamental_ransac_batch_dev(lib, batch, cap, d_keys, d_n, d_prev_xy, d_F, d_ok, d_stats=None, pre_have=None, pre_boxes=None, pre_nboxes=None, max_boxes=0,
                                 threshold=1.0, confidence=0.99, stream=None):
    lib.check(lib.dll.sgx_fundamental_ransac_batch_dev(batch, cap, _vp(d_keys), _vp(d_n), _vp(d_prev_xy), _vp(pre_have), _vp(pre_boxes), _vp(pre_nboxes), max_boxes,
                                                       threshold, confidence, _vp(d_F), _vp(d_ok), _vp(d_stats), _vp(stream)), 'sgx_fundamental_ransac_batch_dev')
