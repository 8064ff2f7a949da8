// sgx_host.hpp — C++ host-side mirror of the reference operator interfaces over the C-ABI (include/sgx.h).
//
// Same class names, method names, argument meaning and error behaviour as the reference
// (src/sg-slam/include/{ORBextractor,ORBmatcher,Optimizer}.h), expressed on plain views instead of
// cv::Mat / Frame* so this header has no OpenCV dependency.  INTEGRATION.md shows the three-line glue that
// adapts cv::Mat / Frame to these views inside the reference tree.  Header-only; link with -lsgx.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "../../include/sgx.h"

namespace sgx {

struct GrayView { const uint8_t *data; int rows, cols, step; bool empty() const { return !data || rows <= 0 || cols <= 0; } };

// flattened ORB_SLAM2::Frame: what the matcher / optimiser read and write
struct FrameView {
    int N = 0;
    std::vector<sgx_keypoint> mvKeysUn;     // == mvKeys when the camera has no distortion (Frame.cc:656-660)
    std::vector<uint8_t> mDescriptors;      // N x 32
    std::vector<float> mvuRight, mvDepth;   // -1 when no depth
    std::vector<int32_t> mvpMapPoints;      // index into the owner of the map points (e.g. last frame), -1 = NULL
    std::vector<uint8_t> mvbOutlier;
    float mTcw[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
    // map points held by this frame's keypoints (used when this frame is the "LastFrame" of the matcher)
    std::vector<uint8_t> mpValid;           // mvpMapPoints[i] != NULL
    std::vector<float> mpWorldPos;          // N x 3   MapPoint::GetWorldPos()
    std::vector<int32_t> mpObservations;    // MapPoint::Observations()
    std::vector<uint8_t> mpDescriptor;      // N x 32  MapPoint::GetDescriptor()
};

// flattened local map (Tracking::mvpLocalMapPoints): what isInFrustum / SearchByProjection(F, vpMapPoints, th) read from each MapPoint
struct LocalMapView {
    int N = 0;
    std::vector<float> mWorldPos, mNormalVector;    // N x 3   GetWorldPos(), GetNormal()
    std::vector<float> mfMinDistance, mfMaxDistance;
    std::vector<uint8_t> mDescriptor;               // N x 32  GetDescriptor()
    std::vector<int32_t> nObs;                      // Observations()
    std::vector<uint8_t> skip;                      // isBad() || mnLastFrameSeen == F.mnId (Tracking.cc:1288-1291)
    std::vector<uint8_t> mbTrackInView;             // out: isInFrustum result (for IncreaseVisible)
};

inline void check(int rc, const char *what) { if (rc != SGX_OK) throw std::runtime_error(std::string(what) + ": " + sgx_status_string(rc)); }

// ORB_SLAM2::ORBextractor (ORBextractor.h:45-111)
class ORBextractor {
public:
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int width = 640, int height = 480)
        : nfeatures_(nfeatures), scaleFactor_(scaleFactor), nlevels_(nlevels)
    {
        sgx_orb_config c{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height, 1};
        check(sgx_orb_create(&c, &h_), "sgx_orb_create");
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        check(sgx_orb_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(), nullptr), "sgx_orb_get_tables");
    }
    ~ORBextractor() { sgx_orb_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // operator()(image, mask, keypoints, descriptors): mask is ignored, as in the reference (Frame.cc:277 passes cv::Mat())
    void operator()(const GrayView &image, const void * /*mask*/, std::vector<sgx_keypoint> &keypoints, std::vector<uint8_t> &descriptors)
    {
        keypoints.clear(); descriptors.clear();
        if (image.empty()) return;                                   // ORBextractor.cc:1048
        const int cap = sgx_orb_keypoint_capacity(h_);
        keypoints.resize(cap); descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(sgx_orb_extract(h_, image.data, image.step, keypoints.data(), descriptors.data(), cap, &n), "sgx_orb_extract");
        keypoints.resize(n); descriptors.resize((size_t)n * 32);
    }
    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return scaleFactor_; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }
    int GetnFeatures() { return nfeatures_; }
    sgx_orb *handle() { return h_; }

private:
    sgx_orb *h_ = nullptr;
    int nfeatures_; float scaleFactor_; int nlevels_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

// ORB_SLAM2::ORBmatcher (ORBmatcher.h:41-89) — SearchByProjection(CurrentFrame, LastFrame, th, bMono)
class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
    ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    static int DescriptorDistance(const uint8_t *a, const uint8_t *b)
    {
        int d = 0;
        for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
        return d;
    }

    // fills CurrentFrame.mvpMapPoints with indices of LastFrame map points; returns nmatches (ORBmatcher.cc:1332-1472)
    int SearchByProjection(FrameView &CurrentFrame, const FrameView &LastFrame, float th, bool bMono,
                           const sgx_camera &cam, const std::vector<float> &scaleFactors)
    {
        CurrentFrame.mvpMapPoints.assign(CurrentFrame.N, -1);
        std::vector<uint8_t> no_outlier(LastFrame.N, 0);
        const uint8_t *outl = LastFrame.mvbOutlier.size() == (size_t)LastFrame.N ? LastFrame.mvbOutlier.data() : no_outlier.data();
        int32_t n = 0;
        check(sgx_match_project_frame(CurrentFrame.N, CurrentFrame.mvKeysUn.data(), CurrentFrame.mDescriptors.data(), CurrentFrame.mvuRight.data(), CurrentFrame.mTcw,
                                      LastFrame.N, LastFrame.mvKeysUn.data(), LastFrame.mpValid.data(), outl, LastFrame.mpWorldPos.data(),
                                      LastFrame.mpObservations.data(), LastFrame.mpDescriptor.data(), LastFrame.mTcw,
                                      &cam, scaleFactors.data(), (int)scaleFactors.size(), th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                      CurrentFrame.mvpMapPoints.data(), &n), "sgx_match_project_frame");
        return n;
    }

    // Tracking::SearchLocalPoints' body (Tracking.cc:1284-1311): isInFrustum(pMP, 0.5) for every local point, then SearchByProjection(F, vpMapPoints, th)
    // (ORBmatcher.cc:45-129).  matched[k] = local point newly assigned to keypoint k of F (or -1); F.mvpMapPoints (>= 0 = holds a point) is only read.
    int SearchByProjection(FrameView &F, LocalMapView &M, float th, const sgx_camera &cam, const std::vector<float> &scaleFactors, std::vector<int32_t> &matched,
                           const std::vector<int32_t> *heldObservations = nullptr)
    {
        matched.assign(F.N, -1); M.mbTrackInView.assign(M.N, 0);
        int32_t n = 0;
        check(sgx_match_project_local(F.N, F.mvKeysUn.data(), F.mDescriptors.data(), F.mvuRight.data(), F.mTcw, heldObservations ? heldObservations->data() : nullptr,
                                      M.N, M.mWorldPos.data(), M.mNormalVector.data(), M.mfMinDistance.data(), M.mfMaxDistance.data(), M.mDescriptor.data(), M.nObs.data(), M.skip.data(),
                                      &cam, scaleFactors.data(), (int)scaleFactors.size(), std::log(scaleFactors[1]), th, mfNNratio, 0.5f,
                                      matched.data(), &n, M.mbTrackInView.data()), "sgx_match_project_local");
        return n;
    }

    // ---- LocalMapping / relocalisation gates (tier N2).  KeyFrameView: the flattened keyframe fields the reference reads.
    struct KeyFrameView {
        int N = 0;
        std::vector<sgx_keypoint> mvKeysUn; std::vector<uint8_t> mDescriptors; std::vector<float> mvuRight;
        std::vector<uint8_t> hasMapPoint;        // GetMapPoint(i) != NULL (for SearchByBoW: ... && !isBad())
        std::vector<int32_t> featNode;           // key of mFeatVec under which keypoint i sits (-1: none)
        float Tcw[16]; float Ow[3];
    };
    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs, const bool bOnlyStereo) (ORBmatcher.cc:659-827)
    int SearchForTriangulation(const KeyFrameView &KF1, const KeyFrameView &KF2, const float F12[9], const sgx_camera &cam2, const std::vector<float> &scaleFactors2,
                               const std::vector<float> &levelSigma2_2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, bool bOnlyStereo)
    {
        std::vector<int32_t> pairs((size_t)2 * (KF1.N > 0 ? KF1.N : 1)); int32_t np = 0;
        check(sgx_match_search_for_triangulation(KF1.N, KF1.mvKeysUn.data(), KF1.mDescriptors.data(), KF1.mvuRight.data(), KF1.hasMapPoint.data(), KF1.featNode.data(), KF1.Ow,
                                                 KF2.N, KF2.mvKeysUn.data(), KF2.mDescriptors.data(), KF2.mvuRight.data(), KF2.hasMapPoint.data(), KF2.featNode.data(), KF2.Tcw,
                                                 F12, &cam2, scaleFactors2.data(), levelSigma2_2.data(), (int)scaleFactors2.size(), bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                                 pairs.data(), &np), "sgx_match_search_for_triangulation");
        vMatchedPairs.clear();
        for (int i = 0; i < np; i++) vMatchedPairs.emplace_back((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]);
        return np;
    }
    // int SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) (ORBmatcher.cc:159-290): matches[j] = keyframe keypoint whose map point keypoint j of F receives
    int SearchByBoW(const KeyFrameView &KF, const FrameView &F, const std::vector<int32_t> &featNodeF, std::vector<int32_t> &matches)
    {
        matches.assign((size_t)(F.N > 0 ? F.N : 1), -1); int32_t n = 0;
        check(sgx_match_search_by_bow(KF.N, KF.mvKeysUn.data(), KF.mDescriptors.data(), KF.hasMapPoint.data(), KF.featNode.data(),
                                      F.N, F.mvKeysUn.data(), F.mDescriptors.data(), featNodeF.data(), mfNNratio, mbCheckOrientation ? 1 : 0, matches.data(), &n), "sgx_match_search_by_bow");
        matches.resize((size_t)F.N);
        return n;
    }
    // the search of int Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th) (ORBmatcher.cc:829-979): bestIdx[i] = keypoint of pKF map point i fuses with (-1: none)
    int Fuse(const KeyFrameView &KF, LocalMapView &M, float th, const sgx_camera &cam, const std::vector<float> &scaleFactors, const std::vector<float> &invLevelSigma2,
             std::vector<int32_t> &bestIdx, std::vector<int32_t> &bestDist)
    {
        bestIdx.assign((size_t)(M.N > 0 ? M.N : 1), -1); bestDist.assign((size_t)(M.N > 0 ? M.N : 1), 256); int32_t n = 0;
        check(sgx_match_fuse_search(KF.N, KF.mvKeysUn.data(), KF.mDescriptors.data(), KF.mvuRight.data(), KF.Tcw, M.N, M.mWorldPos.data(), M.mNormalVector.data(), M.mfMinDistance.data(),
                                    M.mfMaxDistance.data(), M.mDescriptor.data(), M.skip.data(), &cam, scaleFactors.data(), invLevelSigma2.data(), (int)scaleFactors.size(),
                                    std::log(scaleFactors[1]), th, bestIdx.data(), bestDist.data(), &n), "sgx_match_fuse_search");
        bestIdx.resize((size_t)M.N); bestDist.resize((size_t)M.N);
        return n;
    }
    // int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist) (ORBmatcher.cc:1474-1601): M = pKF's map points
    // (M.skip[i] = NULL / isBad() / already found), kfKeysUn = pKF->mvKeysUn; CurrentFrame.mvpMapPoints (>= 0 = holds a point) is read; matched[k] = index into M or -1
    int SearchByProjection(FrameView &CurrentFrame, const std::vector<sgx_keypoint> &kfKeysUn, LocalMapView &M, float th, int ORBdist, const sgx_camera &cam,
                           const std::vector<float> &scaleFactors, std::vector<int32_t> &matched)
    {
        std::vector<uint8_t> held((size_t)(CurrentFrame.N > 0 ? CurrentFrame.N : 1), 0), ok((size_t)(M.N > 0 ? M.N : 1), 0);
        for (int k = 0; k < CurrentFrame.N; k++) held[(size_t)k] = CurrentFrame.mvpMapPoints.size() == (size_t)CurrentFrame.N && CurrentFrame.mvpMapPoints[(size_t)k] >= 0;
        for (int i = 0; i < M.N; i++) ok[(size_t)i] = !M.skip[(size_t)i];
        matched.assign((size_t)(CurrentFrame.N > 0 ? CurrentFrame.N : 1), -1); int32_t n = 0;
        check(sgx_match_project_keyframe(CurrentFrame.N, CurrentFrame.mvKeysUn.data(), CurrentFrame.mDescriptors.data(), held.data(), CurrentFrame.mTcw,
                                         M.N, kfKeysUn.data(), ok.data(), M.mWorldPos.data(), M.mfMinDistance.data(), M.mfMaxDistance.data(), M.mDescriptor.data(),
                                         &cam, scaleFactors.data(), (int)scaleFactors.size(), std::log(scaleFactors[1]), th, ORBdist, mbCheckOrientation ? 1 : 0, matched.data(), &n),
              "sgx_match_project_keyframe");
        matched.resize((size_t)CurrentFrame.N);
        return n;
    }

    // ---- LoopClosing gates (tier N2).  M = candidate map points (M.skip[i] = isBad() / already found / already in pKF); Scw = 4x4 row-major (sRcw | tcw).
    // int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) (ORBmatcher.cc:524-655): hasMapPoint = holds a map point that is not bad;
    // vpMatches12[i1] = keypoint of pKF2 whose map point keypoint i1 of pKF1 is matched with (-1: NULL)
    int SearchByBoW(const KeyFrameView &KF1, const KeyFrameView &KF2, std::vector<int32_t> &vpMatches12)
    {
        vpMatches12.assign((size_t)(KF1.N > 0 ? KF1.N : 1), -1); int32_t n = 0;
        check(sgx_match_search_by_bow_kf(KF1.N, KF1.mvKeysUn.data(), KF1.mDescriptors.data(), KF1.hasMapPoint.data(), KF1.featNode.data(),
                                         KF2.N, KF2.mvKeysUn.data(), KF2.mDescriptors.data(), KF2.hasMapPoint.data(), KF2.featNode.data(), mfNNratio, mbCheckOrientation ? 1 : 0,
                                         vpMatches12.data(), &n), "sgx_match_search_by_bow_kf");
        vpMatches12.resize((size_t)KF1.N);
        return n;
    }
    // int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th)
    // (ORBmatcher.cc:1106-1330).  M1 / M2 = the keyframes' own map points, indexed by keypoint (skip = NULL or bad); vpMatches12 as in include/sgx.h (-1 / >= 0 / -2)
    int SearchBySim3(const KeyFrameView &KF1, LocalMapView &M1, const KeyFrameView &KF2, LocalMapView &M2, std::vector<int32_t> &vpMatches12, float s12, const float R12[9],
                     const float t12[3], float th, const sgx_camera &cam, const std::vector<float> &scaleFactors)
    {
        std::vector<uint8_t> ok1((size_t)(KF1.N > 0 ? KF1.N : 1), 0), ok2((size_t)(KF2.N > 0 ? KF2.N : 1), 0);
        for (int i = 0; i < KF1.N; i++) ok1[(size_t)i] = !M1.skip[(size_t)i];
        for (int i = 0; i < KF2.N; i++) ok2[(size_t)i] = !M2.skip[(size_t)i];
        vpMatches12.resize((size_t)(KF1.N > 0 ? KF1.N : 1), -1); int32_t n = 0;
        check(sgx_match_search_by_sim3(KF1.N, KF1.mvKeysUn.data(), KF1.mDescriptors.data(), KF1.Tcw, ok1.data(), M1.mWorldPos.data(), M1.mfMinDistance.data(), M1.mfMaxDistance.data(), M1.mDescriptor.data(),
                                       KF2.N, KF2.mvKeysUn.data(), KF2.mDescriptors.data(), KF2.Tcw, ok2.data(), M2.mWorldPos.data(), M2.mfMinDistance.data(), M2.mfMaxDistance.data(), M2.mDescriptor.data(),
                                       &cam, scaleFactors.data(), (int)scaleFactors.size(), std::log(scaleFactors[1]), s12, R12, t12, th, vpMatches12.data(), &n), "sgx_match_search_by_sim3");
        vpMatches12.resize((size_t)KF1.N);
        return n;
    }
    // int SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th) (ORBmatcher.cc:292-407):
    // vpMatched[k] >= 0 on entry = slot filled; newly filled slots receive the index into M
    int SearchByProjection(const KeyFrameView &KF, const float Scw[16], LocalMapView &M, std::vector<int32_t> &vpMatched, int th, const sgx_camera &cam, const std::vector<float> &scaleFactors)
    {
        std::vector<uint8_t> held((size_t)(KF.N > 0 ? KF.N : 1), 0);
        for (int k = 0; k < KF.N; k++) held[(size_t)k] = vpMatched.size() == (size_t)KF.N && vpMatched[(size_t)k] >= 0;
        std::vector<int32_t> out((size_t)(KF.N > 0 ? KF.N : 1), -1); int32_t n = 0;
        check(sgx_match_project_sim3(KF.N, KF.mvKeysUn.data(), KF.mDescriptors.data(), held.data(), Scw, M.N, M.mWorldPos.data(), M.mNormalVector.data(), M.mfMinDistance.data(),
                                     M.mfMaxDistance.data(), M.mDescriptor.data(), M.skip.data(), &cam, scaleFactors.data(), (int)scaleFactors.size(), std::log(scaleFactors[1]), th,
                                     out.data(), &n), "sgx_match_project_sim3");
        vpMatched.resize((size_t)KF.N, -1);
        for (int k = 0; k < KF.N; k++) if (out[(size_t)k] >= 0) vpMatched[(size_t)k] = out[(size_t)k];
        return n;
    }
    // the search of int Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint) (ORBmatcher.cc:981-1101)
    int Fuse(const KeyFrameView &KF, const float Scw[16], LocalMapView &M, float th, const sgx_camera &cam, const std::vector<float> &scaleFactors,
             std::vector<int32_t> &bestIdx, std::vector<int32_t> &bestDist)
    {
        bestIdx.assign((size_t)(M.N > 0 ? M.N : 1), -1); bestDist.assign((size_t)(M.N > 0 ? M.N : 1), 256); int32_t n = 0;
        check(sgx_match_fuse_search_sim3(KF.N, KF.mvKeysUn.data(), KF.mDescriptors.data(), Scw, M.N, M.mWorldPos.data(), M.mNormalVector.data(), M.mfMinDistance.data(),
                                         M.mfMaxDistance.data(), M.mDescriptor.data(), M.skip.data(), &cam, scaleFactors.data(), (int)scaleFactors.size(), std::log(scaleFactors[1]), th,
                                         bestIdx.data(), bestDist.data(), &n), "sgx_match_fuse_search_sim3");
        bestIdx.resize((size_t)M.N); bestDist.resize((size_t)M.N);
        return n;
    }

    // int SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize = 10) (ORBmatcher.cc:407-522);
    // vbPrevMatched = N1 x (x, y)
    int SearchForInitialization(const FrameView &F1, const FrameView &F2, std::vector<float> &vbPrevMatched, std::vector<int32_t> &vnMatches12, int windowSize, const sgx_camera &cam)
    {
        vnMatches12.assign((size_t)(F1.N > 0 ? F1.N : 1), -1); int32_t n = 0;
        vbPrevMatched.resize((size_t)2 * (F1.N > 0 ? F1.N : 1));
        check(sgx_match_search_for_initialization(F1.N, F1.mvKeysUn.data(), F1.mDescriptors.data(), F2.N, F2.mvKeysUn.data(), F2.mDescriptors.data(), vbPrevMatched.data(), windowSize,
                                                  mfNNratio, mbCheckOrientation ? 1 : 0, &cam, vnMatches12.data(), &n), "sgx_match_search_for_initialization");
        vnMatches12.resize((size_t)F1.N); vbPrevMatched.resize((size_t)2 * F1.N);
        return n;
    }

protected:
    float mfNNratio; bool mbCheckOrientation;
};

// The numeric steps of LocalMapping / MapPoint that surround the matcher gates and the optimisers, on flattened data
namespace LocalMappingSteps {
    // the per-pair body of LocalMapping::CreateNewMapPoints (LocalMapping.cc:283-421); mvKeys / mvDepth per keyframe; returns nnew, ok[q], x3D[q]
    inline int CreateNewMapPoints(const std::vector<std::pair<size_t, size_t>> &vMatchedIndices,
                                  const ORBmatcher::KeyFrameView &KF1, const std::vector<sgx_keypoint> &mvKeys1, const std::vector<float> &mvDepth1,
                                  const ORBmatcher::KeyFrameView &KF2, const std::vector<sgx_keypoint> &mvKeys2, const std::vector<float> &mvDepth2,
                                  const sgx_camera &cam, const std::vector<float> &scaleFactors, const std::vector<float> &levelSigma2, std::vector<uint8_t> &ok, std::vector<float> &x3D)
    {
        const int np = (int)vMatchedIndices.size();
        std::vector<int32_t> pairs((size_t)2 * (np > 0 ? np : 1));
        for (int i = 0; i < np; i++) { pairs[(size_t)2 * i] = (int32_t)vMatchedIndices[(size_t)i].first; pairs[(size_t)2 * i + 1] = (int32_t)vMatchedIndices[(size_t)i].second; }
        ok.assign((size_t)(np > 0 ? np : 1), 0); x3D.assign((size_t)3 * (np > 0 ? np : 1), 0.f); int32_t nnew = 0;
        check(sgx_triangulate_new_map_points(np, pairs.data(), KF1.N, KF1.mvKeysUn.data(), mvKeys1.data(), KF1.mvuRight.data(), mvDepth1.data(), KF1.Tcw,
                                             KF2.N, KF2.mvKeysUn.data(), mvKeys2.data(), KF2.mvuRight.data(), mvDepth2.data(), KF2.Tcw, &cam, scaleFactors.data(), levelSigma2.data(),
                                             (int)scaleFactors.size(), ok.data(), x3D.data(), &nnew), "sgx_triangulate_new_map_points");
        ok.resize((size_t)np); x3D.resize((size_t)3 * np);
        return nnew;
    }
    // MapPoint::UpdateNormalAndDepth / ComputeDistinctiveDescriptors for a batch of points (MapPoint.cc:330-371, :242-307); observations as CSR in mObservations order
    inline void UpdateNormalAndDepth(const std::vector<float> &xw, const std::vector<int32_t> &obsStart, const std::vector<float> &obsCenter, const std::vector<float> &refCenter,
                                     const std::vector<int32_t> &refLevel, const std::vector<float> &scaleFactors, std::vector<float> &normal, std::vector<float> &minDist, std::vector<float> &maxDist)
    {
        const size_t np_ = refLevel.size();                          // outputs sized here, seeded with the caller's values (points without observations keep them, MapPoint.cc:335-336)
        normal.resize(3 * np_, 0.f); minDist.resize(np_, 0.f); maxDist.resize(np_, 0.f);
        check(sgx_mappoint_update_normal_and_depth((int)refLevel.size(), xw.data(), obsStart.data(), obsCenter.data(), refCenter.data(), refLevel.data(), scaleFactors.data(),
                                                   (int)scaleFactors.size(), normal.data(), minDist.data(), maxDist.data()), "sgx_mappoint_update_normal_and_depth");
    }
    inline void ComputeDistinctiveDescriptors(const std::vector<int32_t> &obsStart, const std::vector<uint8_t> &obsDesc, std::vector<int32_t> &best, std::vector<uint8_t> &mDescriptor)
    {
        const int n = (int)obsStart.size() - 1;
        best.assign((size_t)(n > 0 ? n : 1), -1); mDescriptor.assign((size_t)32 * (n > 0 ? n : 1), 0);
        check(sgx_mappoint_distinctive_descriptors(n, obsStart.data(), obsDesc.data(), best.data(), mDescriptor.data()), "sgx_mappoint_distinctive_descriptors");
        best.resize((size_t)(n > 0 ? n : 0)); mDescriptor.resize((size_t)32 * (n > 0 ? n : 0));
    }
}

// Sim3Solver (src/sg-slam/include/Sim3Solver.h:36-130): constructed from the flattened usable correspondences (what Sim3Solver.cc:40-111 gathers), then the reference's calls
class Sim3Solver {
public:
    // x3dc1 / x3dc2: n x 3 camera-frame points, maxErr1 / maxErr2: 9.210 * mvLevelSigma2[octave], K = (fx, fy, cx, cy), indices1[i] = the i1 of pair i (mvnIndices1), N1 = vpMatched12.size()
    Sim3Solver(const std::vector<float> &x3dc1, const std::vector<float> &x3dc2, const std::vector<float> &maxErr1, const std::vector<float> &maxErr2, const float K1[4], const float K2[4],
               const std::vector<int32_t> &indices1, int N1, bool bFixScale, unsigned randSeed = 0) : idx1_(indices1), N1_(N1), n_((int)maxErr1.size())
    { check(sgx_sim3_solver_create(n_, x3dc1.data(), x3dc2.data(), maxErr1.data(), maxErr2.data(), K1, K2, bFixScale ? 1 : 0, randSeed, &h_), "sgx_sim3_solver_create"); }
    ~Sim3Solver() { if (h_) sgx_sim3_solver_destroy(h_); }
    Sim3Solver(const Sim3Solver &) = delete; Sim3Solver &operator=(const Sim3Solver &) = delete;
    void SetRansacParameters(double probability = 0.99, int minInliers = 6, int maxIterations = 300) { check(sgx_sim3_solver_set_ransac_parameters(h_, probability, minInliers, maxIterations), "sgx_sim3_solver_set_ransac_parameters"); }
    // cv::Mat iterate(int nIterations, bool &bNoMore, vector<bool> &vbInliers, int &nInliers): returns true and fills T12 (4x4 row-major) when a model is found.
    // randDraws (optional): 3 raw rand() values per iteration for callers that share the process-global stream; otherwise the solver's glibc-compatible replica is used.
    bool iterate(int nIterations, bool &bNoMore, std::vector<bool> &vbInliers, int &nInliers, float T12[16], const std::vector<int32_t> *randDraws = nullptr)
    {
        std::vector<uint8_t> inl((size_t)(n_ > 0 ? n_ : 1), 0); int32_t nm = 0, ni = 0, fnd = 0;
        check(sgx_sim3_solver_iterate(h_, nIterations, randDraws ? randDraws->data() : nullptr, T12, &nm, inl.data(), &ni, &fnd, nullptr), "sgx_sim3_solver_iterate");
        bNoMore = nm != 0; nInliers = ni; vbInliers.assign((size_t)N1_, false);
        if (fnd) for (int i = 0; i < n_; i++) if (inl[(size_t)i]) vbInliers[(size_t)idx1_[(size_t)i]] = true;           // vbInliers[mvnIndices1[i]] = true (:196-198)
        return fnd != 0;
    }
    void GetEstimate(float R12[9], float t12[3], float &scale) const { check(sgx_sim3_solver_get_estimate(h_, R12, t12, &scale, nullptr), "sgx_sim3_solver_get_estimate"); }
private:
    sgx_sim3_solver *h_ = nullptr; std::vector<int32_t> idx1_; int N1_, n_;
};

// ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (src/sg-slam/include/ORBVocabulary.h:31-32): the members the reference calls
class ORBVocabulary {
public:
    typedef std::vector<std::pair<int32_t, double>> BowVector;       // DBoW2::BowVector (std::map<WordId, WordValue>) in key order
    ORBVocabulary() {}
    ~ORBVocabulary() { if (h_) sgx_voc_destroy(h_); }
    ORBVocabulary(const ORBVocabulary &) = delete; ORBVocabulary &operator=(const ORBVocabulary &) = delete;
    bool loadFromTextFile(const std::string &filename) { return filename.size() >= 4 && filename.compare(filename.size() - 4, 4, ".txt") == 0 && load(filename); }
    bool loadFromBinaryFile(const std::string &filename) { return !(filename.size() >= 4 && filename.compare(filename.size() - 4, 4, ".txt") == 0) && load(filename); }
    bool empty() const { return size() == 0; }
    unsigned size() const { int32_t nw = 0; if (h_) sgx_voc_info(h_, nullptr, nullptr, nullptr, nullptr, nullptr, &nw); return (unsigned)nw; }
    // void transform(const vector<TDescriptor>& features, BowVector &v, FeatureVector &fv, int levelsup) (TemplatedVocabulary.h:1139-1206); descriptors = N x 32 bytes;
    // featNode[i] = the key of fv under which feature i sits (-1: stopped word) — KeyFrameView::featNode / the featNodeF argument of SearchByBoW
    void transform(const std::vector<uint8_t> &descriptors, BowVector &v, std::vector<int32_t> &featNode, int levelsup) const
    {
        const int n = (int)(descriptors.size() / 32);
        std::vector<int32_t> ids((size_t)(n > 0 ? n : 1)); std::vector<double> w((size_t)(n > 0 ? n : 1)); int32_t nb = 0;
        featNode.assign((size_t)(n > 0 ? n : 1), -1);
        check(sgx_voc_transform(h_, n, descriptors.data(), levelsup, ids.data(), w.data(), &nb, featNode.data(), nullptr), "sgx_voc_transform");
        featNode.resize((size_t)n); v.clear();
        for (int i = 0; i < nb; i++) v.emplace_back(ids[(size_t)i], w[(size_t)i]);
    }
    // double score(const BowVector &a, const BowVector &b) (:1210-1215)
    double score(const BowVector &a, const BowVector &b) const
    {
        std::vector<int32_t> ia, ib; std::vector<double> wa, wb;
        for (auto &e : a) { ia.push_back(e.first); wa.push_back(e.second); }
        for (auto &e : b) { ib.push_back(e.first); wb.push_back(e.second); }
        double s = 0;
        check(sgx_voc_score(h_, (int)ia.size(), ia.data(), wa.data(), (int)ib.size(), ib.data(), wb.data(), &s), "sgx_voc_score");
        return s;
    }
    sgx_voc *handle() const { return h_; }
private:
    bool load(const std::string &f) { sgx_voc *h = nullptr; if (sgx_voc_load(f.c_str(), &h) != SGX_OK) return false; if (h_) sgx_voc_destroy(h_); h_ = h; return true; }
    sgx_voc *h_ = nullptr;
};

// the two OpenCV calls of Frame::RmDynamicPointWithSemanticAndGeometry (Frame.cc:445, :469-472)
class OpticalFlowLK {                                                // cv::calcOpticalFlowPyrLK(cur, prev, pts, nextPts, status, err, Size(21,21), 3, TermCriteria(ITER|EPS, 30, 0.01))
public:
    OpticalFlowLK(int width, int height) : w_(width) { sgx_flow_config c{width, height, 1, 21, 3, 30, 0.01}; check(sgx_flow_create(&c, &h_), "sgx_flow_create"); }
    ~OpticalFlowLK() { if (h_) sgx_flow_destroy(h_); }
    OpticalFlowLK(const OpticalFlowLK &) = delete; OpticalFlowLK &operator=(const OpticalFlowLK &) = delete;
    void operator()(const uint8_t *imFrom, const uint8_t *imTo, const std::vector<float> &pts /* x0 y0 x1 y1 .. */, std::vector<float> &nextPts, std::vector<uint8_t> &status)
    {
        const int n = (int)(pts.size() / 2);
        nextPts.assign(pts.size() ? pts.size() : 2, 0.f); status.assign((size_t)(n > 0 ? n : 1), 0);
        check(sgx_flow_lk(h_, imFrom, imTo, w_, pts.data(), n, nextPts.data(), status.data()), "sgx_flow_lk");
        nextPts.resize(pts.size()); status.resize((size_t)n);
    }
private:
    sgx_flow *h_ = nullptr; int w_;
};
// cv::findFundamentalMat(points1, points2, cv::FM_RANSAC, 1.0, 0.99): returns false for OpenCV's empty Mat
inline bool findFundamentalMat(const std::vector<float> &points1, const std::vector<float> &points2, double F[9], double param1 = 1.0, double param2 = 0.99)
{
    int32_t ok = 0;
    check(sgx_find_fundamental_mat(points1.data(), points2.data(), (int)(points1.size() / 2), param1, param2, F, &ok, nullptr), "sgx_find_fundamental_mat");
    return ok != 0;
}

// ORB_SLAM2::Optimizer (Optimizer.h:40-58) — static int PoseOptimization(Frame *pFrame)
class Optimizer {
public:
    // map points of pFrame->mvpMapPoints[i] live in `owner` (the frame the matcher matched against)
    static int PoseOptimization(FrameView *pFrame, const FrameView &owner, const sgx_camera &cam, const std::vector<float> &invLevelSigma2)
    {
        const int N = pFrame->N;
        std::vector<uint8_t> has(N, 0); std::vector<float> xw((size_t)N * 3, 0.f);
        for (int i = 0; i < N; i++) {
            const int m = pFrame->mvpMapPoints[i];
            if (m >= 0) { has[i] = 1; std::memcpy(&xw[3 * (size_t)i], &owner.mpWorldPos[3 * (size_t)m], 12); }
        }
        pFrame->mvbOutlier.assign(N, 0);
        int32_t ninl = 0;
        check(sgx_pose_optimization(N, pFrame->mvKeysUn.data(), pFrame->mvuRight.data(), has.data(), xw.data(), invLevelSigma2.data(),
                                    (int)invLevelSigma2.size(), &cam, pFrame->mTcw, pFrame->mvbOutlier.data(), &ninl), "sgx_pose_optimization");
        return ninl;
    }

    // static void LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap) (Optimizer.cc:453-778) on the flattened local graph the caller
    // collects at Optimizer.cc:455-653: poses / points are updated in place, the return value lists the erased observations.
    struct LocalGraph {
        std::vector<float> poses;            // n_poses x 16 (Tcw)
        std::vector<uint8_t> pose_fixed;     // 0 local keyframe, 1 fixed camera, 2 local keyframe with mnId == 0
        std::vector<float> points;           // n_points x 3
        std::vector<int32_t> edge_pose, edge_point;
        std::vector<float> edge_obs;         // n_edges x 3 (u, v, uR; uR < 0 = monocular)
        std::vector<float> edge_info;        // mvInvLevelSigma2[octave]
    };
    static std::vector<uint8_t> LocalBundleAdjustment(LocalGraph &g, const sgx_camera &cam, const volatile int32_t *pbStopFlag = nullptr, sgx_ba_stats *stats = nullptr)
    {
        sgx_ba_problem P{(int32_t)g.pose_fixed.size(), (int32_t)(g.points.size() / 3), (int32_t)g.edge_pose.size(), g.poses.data(), g.pose_fixed.data(), g.points.data(),
                         g.edge_pose.data(), g.edge_point.data(), g.edge_obs.data(), g.edge_info.data()};
        std::vector<uint8_t> erase(g.edge_pose.size(), 0);
        check(sgx_local_bundle_adjustment(&P, &cam, pbStopFlag, erase.data(), stats), "sgx_local_bundle_adjustment");
        return erase;
    }
    // static void BundleAdjustment(const vector<KeyFrame*> &vpKFs, const vector<MapPoint*> &vpMP, int nIterations, bool *pbStopFlag, const unsigned long nLoopKF, const bool bRobust)
    // (Optimizer.cc:49-237) on the flattened graph (pose_fixed != 0 for mnId == 0): poses / points updated in place
    static void BundleAdjustment(LocalGraph &g, const sgx_camera &cam, int nIterations = 5, const volatile int32_t *pbStopFlag = nullptr, bool bRobust = true, sgx_ba_stats *stats = nullptr)
    {
        sgx_ba_problem P{(int32_t)g.pose_fixed.size(), (int32_t)(g.points.size() / 3), (int32_t)g.edge_pose.size(), g.poses.data(), g.pose_fixed.data(), g.points.data(),
                         g.edge_pose.data(), g.edge_point.data(), g.edge_obs.data(), g.edge_info.data()};
        check(sgx_bundle_adjustment(&P, &cam, nIterations, pbStopFlag, bRobust ? 1 : 0, stats), "sgx_bundle_adjustment");
    }
    // g2o::Sim3 as (qx, qy, qz, qw, tx, ty, tz, s)
    struct Sim3 { double v[8]; };
    // static int OptimizeSim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches1, g2o::Sim3 &g2oS12, const float th2, const bool bFixScale) (Optimizer.cc:1046-1257)
    // on the correspondences the reference turns into edge pairs (include/sgx.h); inlier[i] == 0 <=> vpMatches1[idx] = NULL
    struct Sim3Pairs { std::vector<float> p1c, p2c, obs1, obs2, info1, info2; };
    static int OptimizeSim3(const Sim3Pairs &c, const float K1[4], const float K2[4], Sim3 &g2oS12, float th2, bool bFixScale, std::vector<uint8_t> &inlier)
    {
        const int n = (int)c.info1.size();
        inlier.assign((size_t)(n > 0 ? n : 1), 0); int32_t nin = 0;
        check(sgx_optimize_sim3(n, c.p1c.data(), c.p2c.data(), c.obs1.data(), c.obs2.data(), c.info1.data(), c.info2.data(), K1, K2, g2oS12.v, th2, bFixScale ? 1 : 0, inlier.data(), nullptr, &nin),
              "sgx_optimize_sim3");
        inlier.resize((size_t)n);
        return nin;
    }
    // the optimisation of static void OptimizeEssentialGraph(...) (Optimizer.cc:781-1042) on the flattened pose graph; vertices are updated in place
    struct PoseGraph { std::vector<Sim3> vScw; std::vector<uint8_t> fixed; std::vector<int32_t> e_i, e_j; std::vector<Sim3> e_meas; };
    static void OptimizeEssentialGraph(PoseGraph &g, bool bFixScale, int iterations = 20, double stats[3] = nullptr)
    {
        std::vector<Sim3> out(g.vScw.size() ? g.vScw.size() : 1);
        check(sgx_optimize_essential_graph((int)g.vScw.size(), g.vScw.empty() ? nullptr : g.vScw[0].v, g.fixed.data(), (int)g.e_i.size(), g.e_i.data(), g.e_j.data(),
                                           g.e_meas.empty() ? nullptr : g.e_meas[0].v, bFixScale ? 1 : 0, iterations, out[0].v, stats), "sgx_optimize_essential_graph");
        out.resize(g.vScw.size()); g.vScw = out;
    }
};

// ORB_SLAM2::Detector2D (Detector2D.h:45-67): same public result members as the reference (read by Frame.cc:482-500)
struct Object2D { float x, y, w, h; float prob; int id; };          // cv::Rect_<float> rect; float prob; int id (name = class_names[id])
class Detector2D {
public:
    // param_text / bin: the contents of ./Thirdparty/ncnn_model/mobilenetv3_ssdlite_voc.{param,bin} (Detector2D.cc:24-25 reads them from the CWD)
    Detector2D(float detection_confidence_threshold, float dynamic_detection_confidence_threshold, const std::string &param_text, const std::vector<uint8_t> &bin,
               int width = 640, int height = 480)
    {
        check(sgx_det_create(param_text.c_str(), bin.data(), bin.size(), width, height, 1, detection_confidence_threshold, dynamic_detection_confidence_threshold, &h_), "sgx_det_create");
    }
    ~Detector2D() { if (h_) sgx_det_destroy(h_); }
    Detector2D(const Detector2D &) = delete; Detector2D &operator=(const Detector2D &) = delete;

    void detect(const uint8_t *bgr, int step)                       // void detect(const cv::Mat &bgr) (Detector2D.cc:34-89)
    {
        sgx_det_result r;
        check(sgx_det_detect(h_, bgr, step, 1, &r), "sgx_det_detect");
        auto conv = [](const sgx_object2d &o) { return Object2D{o.x, o.y, o.w, o.h, o.prob, o.id}; };
        mvObjects2D.clear(); mvPotentialDynamicBorderForMapping.clear(); mvPotentialDynamicBorderForRmDynamicFeature.clear(); raw.assign(r.raw, r.raw + r.n_raw);
        for (int i = 0; i < r.n_objects; i++) mvObjects2D.push_back(conv(r.objects[i]));
        for (int i = 0; i < r.n_map_boxes; i++) mvPotentialDynamicBorderForMapping.push_back(conv(r.map_boxes[i]));
        for (int i = 0; i < r.n_rm_boxes; i++) mvPotentialDynamicBorderForRmDynamicFeature.push_back(conv(r.rm_boxes[i]));
        mbHaveDynamicObjectForMapping = r.have_dynamic_for_mapping != 0; mbHaveDynamicObjectForRmDynamicFeature = r.have_dynamic_for_rm_feature != 0;
    }
    std::vector<Object2D> mvObjects2D, mvPotentialDynamicBorderForMapping, mvPotentialDynamicBorderForRmDynamicFeature;
    bool mbHaveDynamicObjectForMapping = false, mbHaveDynamicObjectForRmDynamicFeature = false;
    std::vector<sgx_detection> raw;                                  // ncnn detection_out rows (test tap)
private:
    sgx_det *h_ = nullptr;
public:
    sgx_det *handle() { return h_; }
};

// The pipelined per-frame host (sgx_tracker_*): S RGB-D streams tracked in lock-step with Tracking::GrabImageRGBD's call order (src/sg-slam/src/Tracking.cc:206-251,
// :906-1013) on three event-chained HIP streams inside the library.  GrabImagesRGBD(slot) = one frame of every stream from the pinned staging buffers of `slot`
// (imRGB as cv::imread delivers it — interleaved 8-bit BGR — and the raw 16-bit depth map, rgbd_tum.cc:114-115), asynchronous; Pose() synchronises.
class TrackingPipeline {
public:
    TrackingPipeline(int streams, const sgx_camera &cam, float depthMapFactor, Detector2D *detector = nullptr, int width = 640, int height = 480, int nFeatures = 1000,
                     float scaleFactor = 1.2f, int nLevels = 8, int iniThFAST = 20, int minThFAST = 7, bool localMap = true, bool dynamicMask = true, int maxBoxes = 8)
        : S_(streams), W_(width), H_(height)
    {
        sgx_tracker_config c; std::memset(&c, 0, sizeof c);
        c.streams = streams; c.width = width; c.height = height; c.nfeatures = nFeatures; c.scale_factor = scaleFactor; c.nlevels = nLevels; c.ini_th_fast = iniThFAST; c.min_th_fast = minThFAST;
        c.cam = cam; c.depth_map_factor = depthMapFactor; c.th_projection = 15.f; c.local_map = localMap; c.dynamic_mask = dynamicMask; c.max_boxes = maxBoxes; c.pipelined = 1;
        check(sgx_tracker_create(&c, detector ? detector->handle() : nullptr, &h_), "sgx_tracker_create");
    }
    ~TrackingPipeline() { if (h_) sgx_tracker_destroy(h_); }
    TrackingPipeline(const TrackingPipeline &) = delete;
    TrackingPipeline &operator=(const TrackingPipeline &) = delete;
    void SetInitialPose(const std::vector<float> &Tcw) { if ((int)Tcw.size() != 16 * S_) throw std::invalid_argument("SetInitialPose: streams x 16 floats"); check(sgx_tracker_set_initial_pose(h_, Tcw.data()), "sgx_tracker_set_initial_pose"); }
    // staging buffers of slot 0 / 1: bgr = streams x height rows of `pitch` bytes (3 * width used), depth = streams x height x width uint16
    void HostBuffers(int slot, uint8_t **bgr, int *pitch, uint16_t **depth) { check(sgx_tracker_host_buffers(h_, slot, bgr, pitch, depth), "sgx_tracker_host_buffers"); }
    void GrabImagesRGBD(int slot, bool rgbOrder = true) { check(sgx_tracker_step_host(h_, slot, rgbOrder ? 1 : 0), "sgx_tracker_step_host"); }
    void Synchronize() { check(sgx_tracker_sync(h_), "sgx_tracker_sync"); }
    // blocks until the device has read the input images of the step issued stepsBack (0..2) calls ago; HostBuffers(slot) does the same for a pinned staging slot
    void WaitInputs(int stepsBack = 0) { check(sgx_tracker_wait_inputs(h_, stepsBack), "sgx_tracker_wait_inputs"); }
    // mCurrentFrame.mTcw of every stream (streams x 16) and the tracking counts of the frame tracked last (synchronises)
    void Pose(std::vector<float> &Tcw, std::vector<int32_t> *nKeys = nullptr, std::vector<int32_t> *nMatches = nullptr, std::vector<int32_t> *nInliers = nullptr)
    {
        Tcw.resize((size_t)16 * S_);
        if (nKeys) nKeys->resize(S_);
        if (nMatches) nMatches->resize(S_);
        if (nInliers) nInliers->resize(S_);
        check(sgx_tracker_read(h_, Tcw.data(), nKeys ? nKeys->data() : nullptr, nMatches ? nMatches->data() : nullptr, nullptr, nullptr, nInliers ? nInliers->data() : nullptr, nullptr, nullptr, nullptr),
              "sgx_tracker_read");
    }
    int streams() const { return S_; }
    sgx_tracker *handle() { return h_; }
private:
    sgx_tracker *h_ = nullptr; int S_, W_, H_;
};

}  // namespace sgx
