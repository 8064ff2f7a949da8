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


def TriangulateNewMapPoints(pairs, kf1, kf2, cam, scale_factors, level_sigma2, lib=None):
    """the per-pair body of LocalMapping::CreateNewMapPoints (LocalMapping.cc:283-421).  kf*: keys_un, keys (mvKeys; defaults to keys_un), uright, depth, Tcw.
    Returns (nnew, ok[npairs], x3d[npairs, 3])."""
    import ctypes as C
    from .matcher import camera_struct
    lib = lib or load()
    pr = np.ascontiguousarray(pairs, 'i4').reshape(-1, 2); npairs = len(pr)
    def flat(k):
        ku = np.ascontiguousarray(k['keys_un']); kd = np.ascontiguousarray(k.get('keys', k['keys_un']))
        return len(ku), ku, kd, np.ascontiguousarray(k['uright'], 'f4'), np.ascontiguousarray(k['depth'], 'f4'), np.ascontiguousarray(k['Tcw'], 'f4').reshape(16)
    n1, a0, a1, a2, a3, a4 = flat(kf1); n2, b0, b1, b2, b3, b4 = flat(kf2)
    sf = np.ascontiguousarray(scale_factors, 'f4'); sg = np.ascontiguousarray(level_sigma2, 'f4'); cs = camera_struct(cam)
    ok = np.zeros(max(npairs, 1), np.uint8); x = np.zeros((max(npairs, 1), 3), 'f4'); n = np.zeros(1, 'i4')
    lib.check(lib.dll.sgx_triangulate_new_map_points(npairs, _vp(pr), n1, _vp(a0), _vp(a1), _vp(a2), _vp(a3), _vp(a4), n2, _vp(b0), _vp(b1), _vp(b2), _vp(b3), _vp(b4),
                                                     C.byref(cs), _vp(sf), _vp(sg), len(sf), _vp(ok), _vp(x), _vp(n)), 'sgx_triangulate_new_map_points')
    return int(n[0]), ok[:npairs].astype(bool), x[:npairs].copy()
