"""Sim3Solver — Python mirror of src/sg-slam/src/Sim3Solver.cc over the C ABI (the RANSAC initialiser of LoopClosing::ComputeSim3, LoopClosing.cc:274-301)."""
import ctypes as C
import numpy as np
from . import load
from .capi import _vp


class Sim3Solver:
    def __init__(self, x3dc1, x3dc2, max_err1, max_err2, K1, K2, bFixScale=True, rand_seed=0, lib=None):
        """the n usable correspondences as the constructor flattens them (Sim3Solver.cc:40-111): camera-frame points, 9.210 * sigma^2 bounds, K = (fx, fy, cx, cy)"""
        self.lib = lib or load()
        a = np.ascontiguousarray(x3dc1, 'f4').reshape(-1, 3); b = np.ascontiguousarray(x3dc2, 'f4').reshape(-1, 3)
        e1 = np.ascontiguousarray(max_err1, 'f4'); e2 = np.ascontiguousarray(max_err2, 'f4'); k1 = np.ascontiguousarray(K1, 'f4'); k2 = np.ascontiguousarray(K2, 'f4')
        self.N = len(a); self.h = C.c_void_p()
        self.lib.check(self.lib.dll.sgx_sim3_solver_create(self.N, _vp(a), _vp(b), _vp(e1), _vp(e2), _vp(k1), _vp(k2), int(bool(bFixScale)), int(rand_seed), C.byref(self.h)), 'sgx_sim3_solver_create')

    def SetRansacParameters(self, probability=0.99, minInliers=6, maxIterations=300):
        self.lib.check(self.lib.dll.sgx_sim3_solver_set_ransac_parameters(self.h, float(probability), int(minInliers), int(maxIterations)), 'sgx_sim3_solver_set_ransac_parameters')

    def iterate(self, nIterations, rand_draws=None):
        """(T12 or None, bNoMore, vbInliers[n], nInliers, iterations_run)"""
        T = np.zeros(16, 'f4'); nm = C.c_int32(); inl = np.zeros(max(self.N, 1), np.uint8); ni = C.c_int32(); fnd = C.c_int32(); run = C.c_int32()
        d = np.ascontiguousarray(rand_draws, 'i4') if rand_draws is not None else None
        self.lib.check(self.lib.dll.sgx_sim3_solver_iterate(self.h, int(nIterations), _vp(d) if d is not None else None, _vp(T), C.byref(nm), _vp(inl), C.byref(ni), C.byref(fnd), C.byref(run)),
                       'sgx_sim3_solver_iterate')
        return (T.reshape(4, 4) if fnd.value else None), bool(nm.value), inl[:self.N].astype(bool), int(ni.value), int(run.value)

    def find(self, rand_draws=None):
        return self.iterate(self.max_iterations(), rand_draws)

    def estimate(self):
        R = np.zeros(9, 'f4'); t = np.zeros(3, 'f4'); s = C.c_float(); m = C.c_int32()
        self.lib.check(self.lib.dll.sgx_sim3_solver_get_estimate(self.h, _vp(R), _vp(t), C.byref(s), C.byref(m)), 'sgx_sim3_solver_get_estimate')
        return R.reshape(3, 3), t, float(s.value)

    def max_iterations(self):
        m = C.c_int32(); self.lib.check(self.lib.dll.sgx_sim3_solver_get_estimate(self.h, None, None, None, C.byref(m)), 'sgx_sim3_solver_get_estimate'); return int(m.value)

    def close(self):
        if self.h: self.lib.dll.sgx_sim3_solver_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass
