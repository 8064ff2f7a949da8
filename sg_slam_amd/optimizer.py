"""Optimizer — host-side mirror of ORB_SLAM2::Optimizer (reference: src/sg-slam/include/Optimizer.h:40-58)
over the C-ABI.  Implemented on the device so far:
  PoseOptimization(pFrame)     src/sg-slam/src/Optimizer.cc:239-451
The Frame is a flattened dict: keys (mvKeysUn, KP_DTYPE), uright, has_mp, xw (map-point world
positions per keypoint), Tcw (4x4 f32).  PoseOptimization updates frame['Tcw'] and frame['outlier']
in place and returns the inlier count, like the reference mutates pFrame."""
import ctypes as C
import numpy as np
from .capi import _vp, BaProblem, BaStats
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

    @staticmethod
    def LocalBundleAdjustment(problem, cam, stop_flag=None, lib=None):
        """Optimizer::LocalBundleAdjustment on a flattened local graph (see include/sgx.h sgx_ba_problem).
        problem: dict(poses[np,4,4] f32, pose_fixed[np] u8, points[nl,3] f32, edge_pose, edge_point [ne] i4,
        edge_obs[ne,3] f32, edge_info[ne] f32).  Updates problem['poses'] / ['points'] in place (like the reference
        mutates KeyFrames / MapPoints) and returns (erase[ne] u8, stats)."""
        lib = lib if lib is not None else load()
        poses = np.ascontiguousarray(problem['poses'], 'f4').reshape(-1, 16).copy(); fixed = np.ascontiguousarray(problem['pose_fixed'], np.uint8)
        pts = np.ascontiguousarray(problem['points'], 'f4').copy()
        ep = np.ascontiguousarray(problem['edge_pose'], 'i4'); el = np.ascontiguousarray(problem['edge_point'], 'i4')
        eo = np.ascontiguousarray(problem['edge_obs'], 'f4'); ei = np.ascontiguousarray(problem['edge_info'], 'f4')
        P = BaProblem(len(poses), len(pts), len(ep), poses.ctypes.data, fixed.ctypes.data, pts.ctypes.data, ep.ctypes.data, el.ctypes.data,
                      eo.ctypes.data, ei.ctypes.data)
        erase = np.zeros(len(ep), np.uint8); st = BaStats()
        cs = camera_struct(cam)
        stop = None if stop_flag is None else _vp(stop_flag)
        lib.check(lib.dll.sgx_local_bundle_adjustment(C.byref(P), C.byref(cs), stop, _vp(erase), C.byref(st)), 'sgx_local_bundle_adjustment')
        problem['poses'] = poses.reshape(-1, 4, 4); problem['points'] = pts
        return erase, dict(iterations=(st.iterations_first, st.iterations_second), chi2=(st.chi2_first, st.chi2_second), free_poses=st.free_poses)

    @staticmethod
    def BundleAdjustment(problem, cam, nIterations=5, stop_flag=None, nLoopKF=0, bRobust=True, lib=None):
        """Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust) on the flattened graph (pose_fixed != 0 exactly for the keyframe
        with mnId == 0).  nLoopKF == 0: problem['poses'] / ['points'] are updated in place (SetPose / SetWorldPos); otherwise the results are returned under
        'poses_gba' / 'points_gba' (mTcwGBA / mPosGBA, mnBAGlobalForKF = nLoopKF) and the problem is left untouched.  Returns stats."""
        lib = lib if lib is not None else load()
        poses = np.ascontiguousarray(problem['poses'], 'f4').reshape(-1, 16).copy(); fixed = np.ascontiguousarray(problem['pose_fixed'], np.uint8)
        pts = np.ascontiguousarray(problem['points'], 'f4').copy()
        ep = np.ascontiguousarray(problem['edge_pose'], 'i4'); el = np.ascontiguousarray(problem['edge_point'], 'i4')
        eo = np.ascontiguousarray(problem['edge_obs'], 'f4'); ei = np.ascontiguousarray(problem['edge_info'], 'f4')
        P = BaProblem(len(poses), len(pts), len(ep), poses.ctypes.data, fixed.ctypes.data, pts.ctypes.data, ep.ctypes.data, el.ctypes.data,
                      eo.ctypes.data, ei.ctypes.data)
        st = BaStats(); cs = camera_struct(cam)
        stop = None if stop_flag is None else _vp(stop_flag)
        lib.check(lib.dll.sgx_bundle_adjustment(C.byref(P), C.byref(cs), int(nIterations), stop, int(bool(bRobust)), C.byref(st)), 'sgx_bundle_adjustment')
        if nLoopKF == 0:
            problem['poses'] = poses.reshape(-1, 4, 4); problem['points'] = pts
        else:
            problem['poses_gba'] = poses.reshape(-1, 4, 4); problem['points_gba'] = pts; problem['mnBAGlobalForKF'] = nLoopKF
        return dict(iterations=st.iterations_first, chi2=st.chi2_first, free_poses=st.free_poses)

    @staticmethod
    def OptimizeSim3(p1c, p2c, obs1, obs2, info1, info2, K1, K2, S12, th2=10.0, bFixScale=False, lib=None):
        """Optimizer::OptimizeSim3(pKF1, pKF2, vpMatches1, g2oS12, th2, bFixScale) (Optimizer.cc:1046-1257) on the flattened correspondences (see include/sgx.h):
        returns (nIn, S12[8] = (qx, qy, qz, qw, tx, ty, tz, s), inlier[n], iterations[2])."""
        lib = lib if lib is not None else load()
        p1c = np.ascontiguousarray(p1c, 'f4').reshape(-1, 3); p2c = np.ascontiguousarray(p2c, 'f4').reshape(-1, 3); n = len(p1c)
        o1 = np.ascontiguousarray(obs1, 'f4').reshape(-1, 2); o2 = np.ascontiguousarray(obs2, 'f4').reshape(-1, 2)
        i1 = np.ascontiguousarray(info1, 'f4'); i2 = np.ascontiguousarray(info2, 'f4')
        k1 = np.ascontiguousarray(K1, 'f4'); k2 = np.ascontiguousarray(K2, 'f4')
        S = np.ascontiguousarray(S12, 'f8').copy(); inl = np.zeros(max(n, 1), np.uint8); it = np.zeros(2, 'i4'); nin = np.zeros(1, 'i4')
        lib.check(lib.dll.sgx_optimize_sim3(n, _vp(p1c), _vp(p2c), _vp(o1), _vp(o2), _vp(i1), _vp(i2), _vp(k1), _vp(k2), _vp(S), float(th2), int(bool(bFixScale)), _vp(inl), _vp(it), _vp(nin)),
                  'sgx_optimize_sim3')
        return int(nin[0]), S, inl[:n].copy(), it

    @staticmethod
    def OptimizeEssentialGraph(S, fixed, e_i, e_j, e_meas, bFixScale=True, iterations=20, lib=None):
        """the optimisation of Optimizer::OptimizeEssentialGraph (Optimizer.cc:781-1042) on a flattened pose graph (see include/sgx.h): (S_out[nv, 8], stats[iterations, chi2 before, after])"""
        lib = lib if lib is not None else load()
        S = np.ascontiguousarray(S, 'f8').reshape(-1, 8); fx = np.ascontiguousarray(fixed, np.uint8)
        ei = np.ascontiguousarray(e_i, 'i4'); ej = np.ascontiguousarray(e_j, 'i4'); em = np.ascontiguousarray(e_meas, 'f8').reshape(-1, 8)
        out = np.zeros_like(S); st = np.zeros(3, 'f8')
        lib.check(lib.dll.sgx_optimize_essential_graph(len(S), _vp(S), _vp(fx), len(ei), _vp(ei), _vp(ej), _vp(em), int(bool(bFixScale)), int(iterations), _vp(out), _vp(st)), 'sgx_optimize_essential_graph')
        return out, st

    @staticmethod
    def CorrectMapPoints(xw, ref, Srw, corrected_Swr, lib=None):
        """Optimizer.cc:1004-1041: P' = correctedSwr.map(Srw.map(P)) for every map point (ref = its reference keyframe's vertex index)"""
        lib = lib if lib is not None else load()
        xw = np.ascontiguousarray(xw, 'f4').reshape(-1, 3); ref = np.ascontiguousarray(ref, 'i4')
        a = np.ascontiguousarray(Srw, 'f8').reshape(-1, 8); c = np.ascontiguousarray(corrected_Swr, 'f8').reshape(-1, 8)
        out = np.zeros_like(xw)
        lib.check(lib.dll.sgx_correct_map_points(len(xw), _vp(xw), _vp(ref), len(a), _vp(a), _vp(c), _vp(out)), 'sgx_correct_map_points')
        return out
