"""Batched MapPoint post-steps over the C ABI: MapPoint::UpdateNormalAndDepth / ComputeDistinctiveDescriptors (src/sg-slam/src/MapPoint.cc:330-371, :242-307) —
what the optimisers and the map-point creation code call per point (Optimizer.cc:227,776,1042, LocalMapping.cc:152-153, Tracking.cc:1233-1234)."""
import numpy as np
from . import load
from .capi import _vp


def UpdateNormalAndDepth(xw, obs_start, obs_center, ref_center, ref_level, scale_factors, normal, min_dist, max_dist, lib=None):
    """n points, observations as CSR in mObservations order; returns updated (normal, min_dist, max_dist) — points without observations keep the inputs"""
    lib = lib or load()
    xw = np.ascontiguousarray(xw, 'f4').reshape(-1, 3); n = len(xw)
    st = np.ascontiguousarray(obs_start, 'i4'); oc = np.ascontiguousarray(obs_center, 'f4').reshape(-1, 3); rc = np.ascontiguousarray(ref_center, 'f4').reshape(-1, 3)
    rl = np.ascontiguousarray(ref_level, 'i4'); sf = np.ascontiguousarray(scale_factors, 'f4')
    nr = np.ascontiguousarray(normal, 'f4').reshape(-1, 3).copy(); mn = np.ascontiguousarray(min_dist, 'f4').copy(); mx = np.ascontiguousarray(max_dist, 'f4').copy()
    lib.check(lib.dll.sgx_mappoint_update_normal_and_depth(n, _vp(xw), _vp(st), _vp(oc), _vp(rc), _vp(rl), _vp(sf), len(sf), _vp(nr), _vp(mn), _vp(mx)), 'sgx_mappoint_update_normal_and_depth')
    return nr, mn, mx


def ComputeDistinctiveDescriptors(obs_start, obs_desc, lib=None):
    """(best index per point within its observation list (-1: none), the chosen descriptor rows)"""
    lib = lib or load()
    st = np.ascontiguousarray(obs_start, 'i4'); n = len(st) - 1; d = np.ascontiguousarray(obs_desc, np.uint8).reshape(-1, 32)
    best = np.full(max(n, 1), -1, 'i4'); out = np.zeros((max(n, 1), 32), np.uint8)
    lib.check(lib.dll.sgx_mappoint_distinctive_descriptors(n, _vp(st), _vp(d), _vp(best), _vp(out)), 'sgx_mappoint_distinctive_descriptors')
    return best[:n].copy(), out[:n].copy()
