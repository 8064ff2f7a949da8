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

    def HammingMatrix(self, desc_a, desc_b):
        """DescriptorDistance of every row pair (na x nb uint16)"""
        a = np.ascontiguousarray(desc_a, np.uint8).reshape(-1, 32); b = np.ascontiguousarray(desc_b, np.uint8).reshape(-1, 32)
        out = np.zeros((len(a), len(b)), np.uint16)
        self.lib.check(self.lib.dll.sgx_hamming_matrix(_vp(a), len(a), _vp(b), len(b), _vp(out)), 'sgx_hamming_matrix')
        return out

    def SearchForTriangulation(self, kf1, kf2, F12, bOnlyStereo, cam2, scale_factors2, level_sigma2_2):
        """ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (ORBmatcher.cc:659-827).  kf*: dicts keys (mvKeysUn), desc, uright, has_mp,
        feat_node (mFeatVec key per keypoint), kf1 additionally cam_center (GetCameraCenter), kf2 Tcw.  Returns (nmatches, vMatchedPairs[n,2])."""
        k1 = np.ascontiguousarray(kf1['keys']); d1 = np.ascontiguousarray(kf1['desc'], np.uint8); u1 = np.ascontiguousarray(kf1['uright'], 'f4')
        h1 = np.ascontiguousarray(kf1['has_mp'], np.uint8); n1 = np.ascontiguousarray(kf1['feat_node'], 'i4'); c1 = np.ascontiguousarray(kf1['cam_center'], 'f4')
        k2 = np.ascontiguousarray(kf2['keys']); d2 = np.ascontiguousarray(kf2['desc'], np.uint8); u2 = np.ascontiguousarray(kf2['uright'], 'f4')
        h2 = np.ascontiguousarray(kf2['has_mp'], np.uint8); n2 = np.ascontiguousarray(kf2['feat_node'], 'i4'); T2 = np.ascontiguousarray(kf2['Tcw'], 'f4').reshape(16)
        F = np.ascontiguousarray(F12, 'f4').reshape(9); sf = np.ascontiguousarray(scale_factors2, 'f4'); sg = np.ascontiguousarray(level_sigma2_2, 'f4')
        pairs = np.zeros((max(len(k1), 1), 2), 'i4'); n = np.zeros(1, 'i4')
        cs = camera_struct(cam2)
        self.lib.check(self.lib.dll.sgx_match_search_for_triangulation(len(k1), _vp(k1), _vp(d1), _vp(u1), _vp(h1), _vp(n1), _vp(c1), len(k2), _vp(k2), _vp(d2), _vp(u2), _vp(h2),
                                                                       _vp(n2), _vp(T2), _vp(F), C.byref(cs), _vp(sf), _vp(sg), len(sf), int(bool(bOnlyStereo)),
                                                                       int(self.mbCheckOrientation), _vp(pairs), _vp(n)), 'sgx_match_search_for_triangulation')
        return int(n[0]), pairs[:n[0]].copy()

    def SearchByBoW(self, kf, F):
        """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (ORBmatcher.cc:159-290).  kf: keys (mvKeysUn), desc, good_mp, feat_node; F: keys, desc, feat_node.
        Returns (nmatches, match_f[nf]): match_f[j] = keyframe keypoint whose map point frame keypoint j gets, or -1."""
        kk = np.ascontiguousarray(kf['keys']); dk = np.ascontiguousarray(kf['desc'], np.uint8); gk = np.ascontiguousarray(kf['good_mp'], np.uint8); nk = np.ascontiguousarray(kf['feat_node'], 'i4')
        kq = np.ascontiguousarray(F['keys']); dq = np.ascontiguousarray(F['desc'], np.uint8); nq = np.ascontiguousarray(F['feat_node'], 'i4')
        match = np.full(max(len(kq), 1), -1, 'i4'); n = np.zeros(1, 'i4')
        self.lib.check(self.lib.dll.sgx_match_search_by_bow(len(kk), _vp(kk), _vp(dk), _vp(gk), _vp(nk), len(kq), _vp(kq), _vp(dq), _vp(nq), float(self.mfNNratio),
                                                            int(self.mbCheckOrientation), _vp(match), _vp(n)), 'sgx_match_search_by_bow')
        return int(n[0]), match[:len(kq)].copy()

    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10, cam=None):
        """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:407-522).  F*: keys (mvKeysUn), desc.
        Returns (nmatches, vnMatches12[n1], vbPrevMatched updated)."""
        a = np.ascontiguousarray(F1['keys']); da = np.ascontiguousarray(F1['desc'], np.uint8); b = np.ascontiguousarray(F2['keys']); db = np.ascontiguousarray(F2['desc'], np.uint8)
        pm = np.ascontiguousarray(vbPrevMatched, 'f4').reshape(-1, 2).copy()
        m = np.full(max(len(a), 1), -1, 'i4'); n = np.zeros(1, 'i4'); cs = camera_struct(cam)
        self.lib.check(self.lib.dll.sgx_match_search_for_initialization(len(a), _vp(a), _vp(da), len(b), _vp(b), _vp(db), _vp(pm), int(windowSize), float(self.mfNNratio),
                                                                        int(self.mbCheckOrientation), C.byref(cs), _vp(m), _vp(n)), 'sgx_match_search_for_initialization')
        return int(n[0]), m[:len(a)].copy(), pm

    def SearchByBoWKF(self, kf1, kf2):
        """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (ORBmatcher.cc:524-655; LoopClosing::ComputeSim3).  kf*: keys, desc, good_mp, feat_node.
        Returns (nmatches, match12[n1]): match12[i1] = keypoint of pKF2 whose map point pKF1's keypoint i1 is matched with, or -1."""
        a = [np.ascontiguousarray(kf1['keys']), np.ascontiguousarray(kf1['desc'], np.uint8), np.ascontiguousarray(kf1['good_mp'], np.uint8), np.ascontiguousarray(kf1['feat_node'], 'i4')]
        b = [np.ascontiguousarray(kf2['keys']), np.ascontiguousarray(kf2['desc'], np.uint8), np.ascontiguousarray(kf2['good_mp'], np.uint8), np.ascontiguousarray(kf2['feat_node'], 'i4')]
        match = np.full(max(len(a[0]), 1), -1, 'i4'); n = np.zeros(1, 'i4')
        self.lib.check(self.lib.dll.sgx_match_search_by_bow_kf(len(a[0]), *[_vp(x) for x in a], len(b[0]), *[_vp(x) for x in b], float(self.mfNNratio),
                                                               int(self.mbCheckOrientation), _vp(match), _vp(n)), 'sgx_match_search_by_bow_kf')
        return int(n[0]), match[:len(a[0])].copy()

    def FuseSearch(self, kf, map_points, th, cam, scale_factors, inv_level_sigma2):
        """the search of ORBmatcher::Fuse(pKF, vpMapPoints, th) (ORBmatcher.cc:829-979): (nFused, best_idx[nm], best_dist[nm]).  kf: keys, desc, uright, Tcw;
        map_points: xw, normal, min_dist, max_dist, desc, skip."""
        k = np.ascontiguousarray(kf['keys']); d = np.ascontiguousarray(kf['desc'], np.uint8); u = np.ascontiguousarray(kf['uright'], 'f4'); T = np.ascontiguousarray(kf['Tcw'], 'f4').reshape(16)
        m = map_points
        xw = np.ascontiguousarray(m['xw'], 'f4'); nr = np.ascontiguousarray(m['normal'], 'f4'); mn = np.ascontiguousarray(m['min_dist'], 'f4'); mx = np.ascontiguousarray(m['max_dist'], 'f4')
        md = np.ascontiguousarray(m['desc'], np.uint8); sk = np.ascontiguousarray(m['skip'], np.uint8)
        sf = np.ascontiguousarray(scale_factors, 'f4'); is2 = np.ascontiguousarray(inv_level_sigma2, 'f4')
        bi = np.full(max(len(xw), 1), -1, 'i4'); bd = np.full(max(len(xw), 1), 256, 'i4'); n = np.zeros(1, 'i4')
        cs = camera_struct(cam)
        self.lib.check(self.lib.dll.sgx_match_fuse_search(len(k), _vp(k), _vp(d), _vp(u), _vp(T), len(xw), _vp(xw), _vp(nr), _vp(mn), _vp(mx), _vp(md), _vp(sk), C.byref(cs), _vp(sf), _vp(is2),
                                                          len(sf), float(np.log(np.float32(sf[1]))), float(th), _vp(bi), _vp(bd), _vp(n)), 'sgx_match_fuse_search')
        return int(n[0]), bi[:len(xw)].copy(), bd[:len(xw)].copy()

    @staticmethod
    def _points(m):
        return [np.ascontiguousarray(m['xw'], 'f4'), np.ascontiguousarray(m['normal'], 'f4'), np.ascontiguousarray(m['min_dist'], 'f4'), np.ascontiguousarray(m['max_dist'], 'f4'),
                np.ascontiguousarray(m['desc'], np.uint8), np.ascontiguousarray(m['skip'], np.uint8)]

    def FuseSearchSim3(self, kf, Scw, map_points, th, cam, scale_factors):
        """the search of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:981-1101): (nFused, best_idx[nm], best_dist[nm]).  kf: keys, desc;
        map_points: xw, normal, min_dist, max_dist, desc, skip."""
        k = np.ascontiguousarray(kf['keys']); d = np.ascontiguousarray(kf['desc'], np.uint8); S = np.ascontiguousarray(Scw, 'f4').reshape(16)
        pts = self._points(map_points); nm = len(pts[0])
        sf = np.ascontiguousarray(scale_factors, 'f4'); cs = camera_struct(cam)
        bi = np.full(max(nm, 1), -1, 'i4'); bd = np.full(max(nm, 1), 256, 'i4'); n = np.zeros(1, 'i4')
        self.lib.check(self.lib.dll.sgx_match_fuse_search_sim3(len(k), _vp(k), _vp(d), _vp(S), nm, *[_vp(x) for x in pts], C.byref(cs), _vp(sf), len(sf),
                                                               float(np.log(np.float32(sf[1]))), float(th), _vp(bi), _vp(bd), _vp(n)), 'sgx_match_fuse_search_sim3')
        return int(n[0]), bi[:nm].copy(), bd[:nm].copy()

    def SearchByProjectionSim3(self, kf, Scw, map_points, th, cam, scale_factors):
        """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:292-407).  kf: keys, desc, matched (vpMatched[k] != NULL on entry).
        Returns (nmatches, matched_out[nk]): matched_out[k] = candidate index newly stored in vpMatched[k], or -1."""
        k = np.ascontiguousarray(kf['keys']); d = np.ascontiguousarray(kf['desc'], np.uint8); mi = np.ascontiguousarray(kf['matched'], np.uint8); S = np.ascontiguousarray(Scw, 'f4').reshape(16)
        pts = self._points(map_points); nm = len(pts[0])
        sf = np.ascontiguousarray(scale_factors, 'f4'); cs = camera_struct(cam)
        mo = np.full(max(len(k), 1), -1, 'i4'); n = np.zeros(1, 'i4')
        self.lib.check(self.lib.dll.sgx_match_project_sim3(len(k), _vp(k), _vp(d), _vp(mi), _vp(S), nm, *[_vp(x) for x in pts], C.byref(cs), _vp(sf), len(sf),
                                                           float(np.log(np.float32(sf[1]))), int(th), _vp(mo), _vp(n)), 'sgx_match_project_sim3')
        return int(n[0]), mo[:len(k)].copy()

    def SearchBySim3(self, kf1, kf2, match12, s12, R12, t12, th, cam, scale_factors):
        """ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:1106-1330).  kf*: keys, desc, Tcw, mp_ok, xw, min_dist, max_dist, mp_desc
        (all per keypoint).  match12: -1 NULL, >= 0 keypoint of pKF2, -2 matched to a point pKF2 does not observe.  Returns (nFound, match12 updated)."""
        def flat(kf):
            return [np.ascontiguousarray(kf['keys']), np.ascontiguousarray(kf['desc'], np.uint8), np.ascontiguousarray(kf['Tcw'], 'f4').reshape(16), np.ascontiguousarray(kf['mp_ok'], np.uint8),
                    np.ascontiguousarray(kf['xw'], 'f4'), np.ascontiguousarray(kf['min_dist'], 'f4'), np.ascontiguousarray(kf['max_dist'], 'f4'), np.ascontiguousarray(kf['mp_desc'], np.uint8)]
        a = flat(kf1); b = flat(kf2)
        sf = np.ascontiguousarray(scale_factors, 'f4'); cs = camera_struct(cam)
        R = np.ascontiguousarray(R12, 'f4').reshape(9); t = np.ascontiguousarray(t12, 'f4').reshape(3)
        m = np.full(max(len(a[0]), 1), -1, 'i4'); m[:len(a[0])] = np.asarray(match12, 'i4'); n = np.zeros(1, 'i4')
        self.lib.check(self.lib.dll.sgx_match_search_by_sim3(len(a[0]), *[_vp(x) for x in a], len(b[0]), *[_vp(x) for x in b], C.byref(cs), _vp(sf), len(sf),
                                                             float(np.log(np.float32(sf[1]))), float(s12), _vp(R), _vp(t), float(th), _vp(m), _vp(n)), 'sgx_match_search_by_sim3')
        return int(n[0]), m[:len(a[0])].copy()

    def SearchByProjectionKF(self, F, kf, th, ORBdist, cam, scale_factors):
        """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1474-1601).  F: keys, desc, has_mp, Tcw; kf: keys, ok (map point present,
        not bad, not already found), xw, min_dist, max_dist, desc.  Returns (nmatches, cur_match[nc]): cur_match[k] = keyframe map point index given to keypoint k, or -1."""
        ck = np.ascontiguousarray(F['keys']); cd = np.ascontiguousarray(F['desc'], np.uint8); ch = np.ascontiguousarray(F['has_mp'], np.uint8); T = np.ascontiguousarray(F['Tcw'], 'f4').reshape(16)
        kk = np.ascontiguousarray(kf['keys']); ok = np.ascontiguousarray(kf['ok'], np.uint8); xw = np.ascontiguousarray(kf['xw'], 'f4')
        mn = np.ascontiguousarray(kf['min_dist'], 'f4'); mx = np.ascontiguousarray(kf['max_dist'], 'f4'); md = np.ascontiguousarray(kf['desc'], np.uint8)
        sf = np.ascontiguousarray(scale_factors, 'f4'); cs = camera_struct(cam)
        match = np.full(max(len(ck), 1), -1, 'i4'); n = np.zeros(1, 'i4')
        self.lib.check(self.lib.dll.sgx_match_project_keyframe(len(ck), _vp(ck), _vp(cd), _vp(ch), _vp(T), len(kk), _vp(kk), _vp(ok), _vp(xw), _vp(mn), _vp(mx), _vp(md), C.byref(cs), _vp(sf), len(sf),
                                                               float(np.log(np.float32(sf[1]))), float(th), int(ORBdist), int(self.mbCheckOrientation), _vp(match), _vp(n)), 'sgx_match_project_keyframe')
        return int(n[0]), match[:len(ck)].copy()
