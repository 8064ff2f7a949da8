/* sgx.h — C ABI of the MI355X-native SG-SLAM tracking hot path (libsgx.so).
 *
 * The reference (silencht/SG-SLAM) has no FFI layer: its Tracking / LocalMapping threads call
 * C++ classes directly.  Each entry point below replaces one of those call sites; the C++
 * adapters in sg_slam_amd/host/ keep the reference class signatures and forward here.
 * All functions return SGX_OK (0) or a negative sgx_status; no C++ types cross this line.
 * Pointers named d_* are DEVICE pointers (HBM); all others are host pointers.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef SGX_H
#define SGX_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum sgx_status {
    SGX_OK = 0,
    SGX_ERR_INVALID = -1,      /* bad argument */
    SGX_ERR_UNSUPPORTED = -2,  /* geometry outside compiled limits */
    SGX_ERR_NOMEM = -3,
    SGX_ERR_DEVICE = -4,       /* HIP runtime error */
    SGX_ERR_OVERFLOW = -5      /* a fixed-capacity device buffer overflowed (result not valid) */
} sgx_status;

const char *sgx_version(void);
const char *sgx_status_string(int status);

/* cv::KeyPoint (OpenCV 3.4 layout, 28 bytes) as produced by ORBextractor::operator()
 * (reference: src/sg-slam/src/ORBextractor.cc:838-848, :1096-1104). */
typedef struct sgx_keypoint {
    float x, y;      /* pt, already multiplied by the level scale for octave > 0 */
    float size;      /* (int)(31 * scale[octave]) */
    float angle;     /* degrees in [0,360), cv::fastAtan2 of the intensity centroid */
    float response;  /* FAST score */
    int32_t octave;
    int32_t class_id; /* -1 */
} sgx_keypoint;

/* ---- ORB extractor ------------------------------------------------------------------------
 * Replaces ORB_SLAM2::ORBextractor (src/sg-slam/include/ORBextractor.h:45-111):
 *   ctor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)   -> sgx_orb_create
 *   operator()(image, mask [ignored], keypoints, descriptors)   -> sgx_orb_extract[_batch_dev]
 *   GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 *   GetInverseScaleSigmaSquares / GetLevels / GetnFeatures         -> sgx_orb_get_tables
 * The image size is fixed per handle (the reference re-derives level sizes per call,
 * ORBextractor.cc:1112-1113; a Tracking instance only ever feeds one size). */
typedef struct sgx_orb_config {
    int32_t nfeatures;      /* ORBextractor.nFeatures  (TUM3.yaml: 1000) */
    float scale_factor;     /* ORBextractor.scaleFactor (1.2) */
    int32_t nlevels;        /* ORBextractor.nLevels (8) */
    int32_t ini_th_fast;    /* ORBextractor.iniThFAST (20) */
    int32_t min_th_fast;    /* ORBextractor.minThFAST (7) */
    int32_t width, height;  /* gray image size (640x480) */
    int32_t max_batch;      /* frames per batched call (device workspace is sized for this) */
} sgx_orb_config;

typedef struct sgx_orb sgx_orb;

int sgx_orb_create(const sgx_orb_config *cfg, sgx_orb **out);
void sgx_orb_destroy(sgx_orb *h);
/* keypoint capacity per frame that callers must provide (sum of per-level quotas + slack;
 * the octree may return up to 3 more than a level's quota, ORBextractor.cc:670-735). */
int sgx_orb_keypoint_capacity(const sgx_orb *h);
/* each array has nlevels entries; any pointer may be NULL */
int sgx_orb_get_tables(const sgx_orb *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                       int32_t *features_per_level);
/* Batched, device-resident.  d_gray: batch frames, each height rows of `pitch` bytes; d_gray and pitch must be multiples of 4
 * (the kernels stage aligned dwords; any image WIDTH is fine — sgx_orb_extract pads its staging copy itself), else SGX_ERR_INVALID.
 * d_kps: batch*cap keypoints, d_desc: batch*cap*32 bytes, d_count: batch int32.
 * Asynchronous on `stream`.  cap must be >= sgx_orb_keypoint_capacity(). */
int sgx_orb_extract_batch_dev(sgx_orb *h, const uint8_t *d_gray, int pitch, int batch,
                              sgx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_count, int cap, void *stream);
/* Single host frame == ORBextractor::operator() (ORBextractor.cc:1045-1106).  Synchronous. */
int sgx_orb_extract(sgx_orb *h, const uint8_t *gray, int stride, sgx_keypoint *kps, uint8_t *desc, int cap, int *n);
/* device-side overflow/status word of the last batched call (synchronises `stream`) */
int sgx_orb_last_status(sgx_orb *h, void *stream);

/* ---- ORB matcher ----------------------------------------------------------------------------
 * Replaces ORB_SLAM2::ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame,
 * const float th, const bool bMono)  (src/sg-slam/include/ORBmatcher.h:56, src/sg-slam/src/ORBmatcher.cc:1332-1472),
 * the motion-model tracker's matcher (Tracking.cc:924-931), including Frame::GetFeaturesInArea's
 * 64x48-grid candidate semantics (Frame.cc:354-419), the greedy "observed map point keeps its
 * keypoint" rule (:1407-1409) and the rotation-histogram filter (:1436-1469).
 * Frames are passed flattened: per keypoint arrays with a common per-frame
 * pitch `cap`.  l_* arrays describe LastFrame.mvpMapPoints[i]: has_mp (non-NULL), outlier
 * (mvbOutlier), xw (GetWorldPos, 3 floats), obs (Observations()), mpdesc (GetDescriptor(), 32 B).
 * cur_match[k] (out) = index i of the last-frame map point assigned to current keypoint k, or -1;
 * the current frame starts with no map points, as at Tracking.cc:919. */
typedef struct sgx_camera {
    float fx, fy, cx, cy, bf;                  /* Frame::fx.. and mbf */
    float min_x, max_x, min_y, max_y;          /* Frame::mnMinX.. (0,cols,0,rows for zero distortion, Frame.cc:707-713) */
} sgx_camera;

int sgx_match_project_frame_batch_dev(
    int batch, int cap,
    const sgx_keypoint *d_ckeys, const uint8_t *d_cdesc, const float *d_curight, const int32_t *d_cn, const float *d_cTcw,
    const sgx_keypoint *d_lkeys, const int32_t *d_ln, const uint8_t *d_l_has_mp, const uint8_t *d_l_outlier, const float *d_l_xw,
    const int32_t *d_l_obs, const uint8_t *d_l_mpdesc, const float *d_lTcw,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float th, int b_mono, int check_orientation,
    int32_t *d_cur_match, int32_t *d_nmatches, void *stream);
/* host pointers, one frame pair, synchronous */
int sgx_match_project_frame(
    int nc, const sgx_keypoint *ckeys, const uint8_t *cdesc, const float *curight, const float *cTcw,
    int nl, const sgx_keypoint *lkeys, const uint8_t *l_has_mp, const uint8_t *l_outlier, const float *l_xw,
    const int32_t *l_obs, const uint8_t *l_mpdesc, const float *lTcw,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float th, int b_mono, int check_orientation,
    int32_t *cur_match, int32_t *nmatches);

/* Tracking::SearchLocalPoints' inner work (src/sg-slam/src/Tracking.cc:1284-1311): Frame::isInFrustum(pMP, viewing_cos_limit = 0.5)
 * (Frame.cc:296-352, MapPoint::PredictScale MapPoint.cc:402-417) for every local map point, then
 * ORBmatcher(nnratio = 0.8).SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, th) (ORBmatcher.cc:45-129).
 * Local map point i: m_xw (GetWorldPos), m_normal (GetNormal), m_min_dist / m_max_dist (mfMinDistance / mfMaxDistance), m_desc,
 * m_obs (Observations()), m_skip (isBad() or mnLastFrameSeen == current frame, Tracking.cc:1288-1291); per-frame pitch mcap (<= 4096).
 * cur_mp_obs[k] = Observations() of the map point keypoint k already holds (-1 = none; NULL = no keypoint holds one).
 * Out: cur_match[k] = local map point newly assigned to keypoint k or -1, nmatches, in_view[i] = mbTrackInView (for IncreaseVisible). */
int sgx_match_project_local_batch_dev(
    int batch, int cap, const sgx_keypoint *d_ckeys, const uint8_t *d_cdesc, const float *d_curight, const int32_t *d_cn, const float *d_cTcw, const int32_t *d_cur_mp_obs,
    int mcap, const int32_t *d_mn, const float *d_m_xw, const float *d_m_normal, const float *d_m_min_dist, const float *d_m_max_dist, const uint8_t *d_m_desc,
    const int32_t *d_m_obs, const uint8_t *d_m_skip,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th, float nnratio, float viewing_cos_limit,
    int32_t *d_cur_match, int32_t *d_nmatches, uint8_t *d_in_view, void *stream);

/* host pointers, one frame, synchronous (cur_mp_obs and in_view may be NULL) */
int sgx_match_project_local(
    int nc, const sgx_keypoint *ckeys, const uint8_t *cdesc, const float *curight, const float *cTcw, const int32_t *cur_mp_obs,
    int nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const int32_t *m_obs, const uint8_t *m_skip,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th, float nnratio, float viewing_cos_limit,
    int32_t *cur_match, int32_t *nmatches, uint8_t *in_view);

/* ---- ORBmatcher gates of the LocalMapping thread (tier N2) --------------------------------------------------------------------
 * static int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b) (src/sg-slam/include/ORBmatcher.h:45, ORBmatcher.cc:1649-1665) for every pair of two
 * descriptor sets (rows of 32 bytes): out[i * nb + j]. */
int sgx_hamming_matrix(const uint8_t *desc_a, int na, const uint8_t *desc_b, int nb, uint16_t *out);
int sgx_hamming_matrix_dev(const uint8_t *d_desc_a, int na, const uint8_t *d_desc_b, int nb, uint16_t *d_out, void *stream);
/* int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs, const bool bOnlyStereo)
 * (src/sg-slam/include/ORBmatcher.h:72-73, src/sg-slam/src/ORBmatcher.cc:659-827; caller LocalMapping::CreateNewMapPoints, LocalMapping.cc:268).  Flattened keyframes:
 * keys*_un = mvKeysUn, desc* = mDescriptors, uright* = mvuRight, has_mp*[i] = GetMapPoint(i) != NULL, feat_node*[i] = the key under which keypoint i sits in mFeatVec
 * (DBoW2 FeatureVector: vocabulary node 4 levels up; -1 = not in the map), cam_center1 = pKF1->GetCameraCenter() (3 floats), Tcw2 = pKF2->GetPose() (4x4 row-major),
 * F12 = 3x3 row-major float, cam2 / scale_factors2 / level_sigma2_2 = pKF2's fx.., mvScaleFactors, mvLevelSigma2.  pairs (out, capacity n1 pairs) = vMatchedPairs as
 * (idx1, idx2) in ascending idx1; *npairs = return value.  Host pointers, synchronous. */
int sgx_match_search_for_triangulation(
    int n1, const sgx_keypoint *keys1_un, const uint8_t *desc1, const float *uright1, const uint8_t *has_mp1, const int32_t *feat_node1, const float *cam_center1,
    int n2, const sgx_keypoint *keys2_un, const uint8_t *desc2, const float *uright2, const uint8_t *has_mp2, const int32_t *feat_node2, const float *Tcw2,
    const float *F12, const sgx_camera *cam2, const float *scale_factors2, const float *level_sigma2_2, int nlevels, int only_stereo, int check_orientation,
    int32_t *pairs, int32_t *npairs);
/* int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) (src/sg-slam/include/ORBmatcher.h:65, src/sg-slam/src/ORBmatcher.cc:159-290;
 * callers Tracking::TrackReferenceKeyFrame Tracking.cc:806, Relocalization :1496): kf_good_mp[i] = the keyframe's keypoint i holds a map point that is not bad,
 * feat_node_* = mFeatVec keys per keypoint (the vocabulary transform itself — DBoW2 + ORBvoc — stays with the caller).  match_f[j] = index of the keyframe keypoint whose
 * map point the frame's keypoint j receives (vpMapPointMatches[j] = vpMapPointsKF[match_f[j]]), -1 = NULL; *nmatches = return value.  Host pointers, synchronous. */
int sgx_match_search_by_bow(
    int nk, const sgx_keypoint *keys_kf_un, const uint8_t *desc_kf, const uint8_t *kf_good_mp, const int32_t *feat_node_kf,
    int nf, const sgx_keypoint *keys_f_un, const uint8_t *desc_f, const int32_t *feat_node_f, float nnratio, int check_orientation,
    int32_t *match_f, int32_t *nmatches);
/* int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) (src/sg-slam/include/ORBmatcher.h:66, src/sg-slam/src/ORBmatcher.cc:524-655;
 * caller LoopClosing::ComputeSim3, LoopClosing.cc:265): good*[i] = keypoint i of that keyframe holds a map point that is not bad; match12[i1] (out, n1 entries) = keypoint of
 * pKF2 whose map point is vpMatches12[i1], -1 = NULL; *nmatches = return value.  Differs from the KeyFrame-Frame overload in the strict `< TH_LOW` gate (:597) and in requiring a
 * map point on both sides (:581-587).  Host pointers, synchronous. */
int sgx_match_search_by_bow_kf(
    int n1, const sgx_keypoint *keys1_un, const uint8_t *desc1, const uint8_t *good1, const int32_t *feat_node1,
    int n2, const sgx_keypoint *keys2_un, const uint8_t *desc2, const uint8_t *good2, const int32_t *feat_node2, float nnratio, int check_orientation,
    int32_t *match12, int32_t *nmatches);
/* The search of int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th = 3.0) (src/sg-slam/include/ORBmatcher.h:80,
 * src/sg-slam/src/ORBmatcher.cc:829-979; caller LocalMapping::SearchInNeighbors, LocalMapping.cc:489,514): for every candidate map point i (m_skip[i] = NULL / isBad() /
 * IsInKeyFrame(pKF); m_min_dist / m_max_dist = mfMinDistance / mfMaxDistance) the keyframe keypoint best_idx[i] it fuses with (-1: none within TH_LOW) and the Hamming
 * distance best_dist[i].  *nfused = return value (number of best_idx >= 0).  The map mutations that follow in the reference (Replace / AddObservation / AddMapPoint,
 * :950-969) are applied by the caller in index order: they do not influence the search.  Host pointers, synchronous. */
int sgx_match_fuse_search(
    int nk, const sgx_keypoint *keys_un, const uint8_t *desc, const float *uright, const float *Tcw,
    int nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
    const sgx_camera *cam, const float *scale_factors, const float *inv_level_sigma2, int nlevels, float log_scale_factor, float th,
    int32_t *best_idx, int32_t *best_dist, int32_t *nfused);

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist)
 * (src/sg-slam/include/ORBmatcher.h:56, src/sg-slam/src/ORBmatcher.cc:1474-1601; caller Tracking::Relocalization, Tracking.cc:1571,1584): projection of the keyframe's map
 * points into the frame.  kf_ok[i] = vpMPs[i] && !isBad() && !sAlreadyFound.count(vpMPs[i]); c_has_mp[k] = CurrentFrame.mvpMapPoints[k] != NULL on entry (such keypoints are
 * never taken); m_min_dist / m_max_dist = mfMinDistance / mfMaxDistance.  cur_match[k] (out) = index i of the keyframe map point keypoint k receives
 * (CurrentFrame.mvpMapPoints[k] = vpMPs[i]), -1 = left alone; *nmatches = return value.  Host pointers, synchronous. */
int sgx_match_project_keyframe(
    int nc, const sgx_keypoint *ckeys_un, const uint8_t *cdesc, const uint8_t *c_has_mp, const float *cTcw,
    int nk, const sgx_keypoint *kf_keys_un, const uint8_t *kf_ok, const float *m_xw, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th, int orb_dist, int check_orientation,
    int32_t *cur_match, int32_t *nmatches);

/* ---- ORBmatcher gates of the LoopClosing thread that project map points through a similarity (tier N2) ----------------------------------------------------------
 * Shared conventions: keys_un / desc = the target keyframe's mvKeysUn / mDescriptors; Scw = 4x4 row-major float (sRcw | tcw), decomposed as the reference does
 * (:301-306, :990-995); m_xw / m_normal / m_min_dist / m_max_dist / m_desc = GetWorldPos, GetNormal, mfMinDistance, mfMaxDistance, GetDescriptor of every candidate;
 * cam supplies fx, fy, cx, cy and the image bounds (KeyFrame::IsInImage), scale_factors = mvScaleFactors, log_scale_factor = mfLogScaleFactor.  Host pointers, synchronous.
 *
 * The search of int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint)
 * (src/sg-slam/include/ORBmatcher.h:83, src/sg-slam/src/ORBmatcher.cc:981-1101; caller LoopClosing::SearchAndFuse, LoopClosing.cc:599): m_skip[i] = isBad() ||
 * pKF->GetMapPoints().count(pMP).  best_idx[i] = keyframe keypoint candidate i lands on (-1: none within TH_LOW), best_dist[i] its Hamming distance (256: none); the caller
 * reads pKF->GetMapPoint(best_idx[i]) to fill vpReplacePoint[i] or to add the observation (:1084-1097).  *nfused = return value. */
int sgx_match_fuse_search_sim3(
    int nk, const sgx_keypoint *keys_un, const uint8_t *desc, const float *Scw,
    int nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th,
    int32_t *best_idx, int32_t *best_dist, int32_t *nfused);
/* int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
 * (src/sg-slam/include/ORBmatcher.h:60, src/sg-slam/src/ORBmatcher.cc:292-407; caller LoopClosing::ComputeSim3, LoopClosing.cc:375): matched_in[k] = vpMatched[k] != NULL on
 * entry, m_skip[i] = isBad() || the point is already in vpMatched.  matched_out[k] = index i of the candidate this call stores in vpMatched[k], -1 = unchanged; the
 * reference's in-order greedy assignment (a point skips keypoints filled by earlier points, :381) is reproduced exactly.  *nmatches = return value. */
int sgx_match_project_sim3(
    int nk, const sgx_keypoint *keys_un, const uint8_t *desc, const uint8_t *matched_in, const float *Scw,
    int nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, int th,
    int32_t *matched_out, int32_t *nmatches);
/* int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th)
 * (src/sg-slam/include/ORBmatcher.h:77, src/sg-slam/src/ORBmatcher.cc:1106-1330; caller LoopClosing::ComputeSim3, LoopClosing.cc:323).  Per keyframe, indexed by keypoint:
 * Tcw = GetPose() (4x4 row-major), mp_ok[i] = GetMapPointMatches()[i] is neither NULL nor bad, m_* = that map point's fields.  R12 = 3x3 row-major, t12 = 3 floats.
 * match12 (in/out, n1 entries): -1 = vpMatches12[i1] is NULL; >= 0 = the keypoint of pKF2 its map point sits on (GetIndexInKeyFrame(pKF2)); -2 = non-NULL but not observed by
 * pKF2.  Pairs that agree in both directions are written as keypoint indices of pKF2 (vpMatches12[i1] = vpMapPoints2[match12[i1]]); *nfound = return value.  Both keyframes
 * share cam / scale_factors (one sensor, one ORB pyramid). */
int sgx_match_search_by_sim3(
    int n1, const sgx_keypoint *keys1_un, const uint8_t *desc1, const float *Tcw1, const uint8_t *mp_ok1, const float *m_xw1, const float *m_min_dist1, const float *m_max_dist1, const uint8_t *m_desc1,
    int n2, const sgx_keypoint *keys2_un, const uint8_t *desc2, const float *Tcw2, const uint8_t *mp_ok2, const float *m_xw2, const float *m_min_dist2, const float *m_max_dist2, const uint8_t *m_desc2,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float s12, const float *R12, const float *t12, float th,
    int32_t *match12, int32_t *nfound);

/* int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize = 10)
 * (src/sg-slam/include/ORBmatcher.h:69, src/sg-slam/src/ORBmatcher.cc:407-522; caller Tracking::MonocularInitialization, Tracking.cc:639, windowSize 100): the monocular
 * initialiser's matcher (the RGB-D system never reaches it; built so that every ORBmatcher method has a device form).  prev_matched (in/out, n1 x 2 floats) = vbPrevMatched,
 * matches12 (out, n1) = vnMatches12; cam supplies the image bounds of F2's grid.  The reference's order-dependent stealing rule (a later keypoint takes a match only with a
 * strictly smaller distance) is reproduced exactly.  Host pointers, synchronous. */
int sgx_match_search_for_initialization(
    int n1, const sgx_keypoint *keys1_un, const uint8_t *desc1, int n2, const sgx_keypoint *keys2_un, const uint8_t *desc2,
    float *prev_matched, int window_size, float nnratio, int check_orientation, const sgx_camera *cam, int32_t *matches12, int32_t *nmatches);

/* The per-pair body of void LocalMapping::CreateNewMapPoints() (src/sg-slam/src/LocalMapping.cc:283-421) for the pairs sgx_match_search_for_triangulation returned for
 * (mpCurrentKeyFrame, pKF2): parallax test (:300-317), linear triangulation by cv::SVD::compute on the 4 x 4 system (:320-338) or the stereo unprojection of the side with
 * the larger parallax (:340-349), positive depth in both keyframes (:353-360), the chi-square reprojection gates (:362-406: 5.991 mono, 7.8 stereo, times mvLevelSigma2) and
 * the scale-consistency gate (:408-423).  keys*_un = mvKeysUn, keys* = mvKeys (KeyFrame::UnprojectStereo reads those, KeyFrame.cc:621-622; the same array when the camera
 * has no distortion), uright* = mvuRight, depth* = mvDepth, Tcw* = GetPose() (4x4 row-major); cam: fx, fy, cx, cy, bf.  ok[q] = 1 and x3d[q] = the new map point of pair q;
 * the map mutations (:425-442: new MapPoint, AddObservation, ComputeDistinctiveDescriptors, UpdateNormalAndDepth ...) stay with the caller.  *nnew = number of ok pairs.
 * Divergence: where the reference would dereference the empty Mat of UnprojectStereo (depth <= 0 on the chosen stereo side) the pair is rejected. */
int sgx_triangulate_new_map_points(
    int npairs, const int32_t *pairs,
    int n1, const sgx_keypoint *keys1_un, const sgx_keypoint *keys1, const float *uright1, const float *depth1, const float *Tcw1,
    int n2, const sgx_keypoint *keys2_un, const sgx_keypoint *keys2, const float *uright2, const float *depth2, const float *Tcw2,
    const sgx_camera *cam, const float *scale_factors, const float *level_sigma2, int nlevels, uint8_t *ok, float *x3d, int32_t *nnew);

/* ---- MapPoint post-steps of the optimisers and of map-point creation, batched over points (host pointers, synchronous) -----------------------------------------------
 * Observations of point p = entries obs_start[p] .. obs_start[p + 1] - 1, IN THE ORDER THE REFERENCE WALKS mObservations (a std::map keyed by KeyFrame*: the float sums
 * below depend on it; the caller flattens in that order).
 * void MapPoint::UpdateNormalAndDepth() (src/sg-slam/src/MapPoint.cc:330-371; callers Optimizer.cc:227,776,1042 — the last statement of BundleAdjustment / LocalBundleAdjustment /
 * OptimizeEssentialGraph per point — LocalMapping.cc:152,444,527, Tracking.cc:572,708,1234): obs_center = GetCameraCenter() of every observing keyframe, ref_center /
 * ref_level = mpRefKF's camera centre and the octave of the point's keypoint in it.  normal / min_dist / max_dist (in/out) = mNormalVector, mfMinDistance, mfMaxDistance;
 * a point without observations keeps its values (:345-346). */
int sgx_mappoint_update_normal_and_depth(int n, const float *xw, const int32_t *obs_start, const float *obs_center, const float *ref_center, const int32_t *ref_level,
                                         const float *scale_factors, int nlevels, float *normal, float *min_dist, float *max_dist);
/* void MapPoint::ComputeDistinctiveDescriptors() (MapPoint.cc:242-307): obs_desc = the observed descriptor rows (keyframes that are not bad).  best[p] = index within the
 * point's list of the descriptor with the least median Hamming distance to the others (first on ties), -1 for an empty list; desc_out (optional, n x 32) = that row. */
int sgx_mappoint_distinctive_descriptors(int n, const int32_t *obs_start, const uint8_t *obs_desc, int32_t *best, uint8_t *desc_out);

/* ---- Sim3Solver (src/sg-slam/include/Sim3Solver.h:36-130, src/sg-slam/src/Sim3Solver.cc), the RANSAC initialiser of LoopClosing::ComputeSim3 (LoopClosing.cc:274-301) ----
 * The constructor's flattening stays with the caller (:40-111): for every usable pair (vpMatched12[i1] set, both map points good and indexed in their keyframes)
 * x3dc1 = Rcw1 * X3D1w + tcw1, x3dc2 = Rcw2 * X3D2w + tcw2 (n x 3 floats), max_err1/2 = 9.210 * mvLevelSigma2[octave] (n floats), K1 / K2 = fx, fy, cx, cy.
 * create runs FromCameraToImage (:407-425) and SetRansacParameters() with the class defaults (0.99, 6, 300).
 * iterate(nIterations, bNoMore, vbInliers, nInliers) (:140-208): every iteration of the call is a hypothesis evaluated on the device (three correspondences, Horn's
 * closed form ComputeSim3 :228-337 in the reference's cv::Mat arithmetic incl. cv::eigen's Jacobi and cv::Rodrigues, CheckInliers :340-365); the sequential accept rule
 * (>= best so far, return at the first model with more than minInliers inliers) is replayed in iteration order, iterations after a success are not consumed.
 * Random numbers: the reference calls the process-global rand() (DUtils::Random::RandomInt, Random.cpp:71-74), three times per iteration.  rand_draws = those raw rand()
 * values (3 x n_iterations; only the first 3 x *iterations_run are consumed — a caller sharing the global stream passes one iteration per call), or NULL: the solver then
 * uses its own replica of glibc's rand() seeded with rand_seed at creation (srand(rand_seed)).
 * Outputs: *found = a model was returned (T12 = 4x4 row-major mBestT12, inliers[n] over the n pairs — the caller maps them through mvnIndices1 — *n_inliers);
 * *no_more = bNoMore.  get_estimate = GetEstimatedRotation / Translation / Scale of the best model so far, and the effective mRansacMaxIts. */
typedef struct sgx_sim3_solver sgx_sim3_solver;
int sgx_sim3_solver_create(int n, const float *x3dc1, const float *x3dc2, const float *max_err1, const float *max_err2, const float *K1, const float *K2, int fix_scale,
                           unsigned rand_seed, sgx_sim3_solver **out);
int sgx_sim3_solver_set_ransac_parameters(sgx_sim3_solver *s, double probability, int min_inliers, int max_iterations);
int sgx_sim3_solver_iterate(sgx_sim3_solver *s, int n_iterations, const int32_t *rand_draws, float *T12, int32_t *no_more, uint8_t *inliers, int32_t *n_inliers,
                            int32_t *found, int32_t *iterations_run);
int sgx_sim3_solver_get_estimate(const sgx_sim3_solver *s, float *R12, float *t12, float *scale, int32_t *max_iterations);
void sgx_sim3_solver_destroy(sgx_sim3_solver *s);

/* ---- ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (src/sg-slam/include/ORBVocabulary.h:31-32) ----------------------------------------------
 * The bag-of-words transform behind Frame::ComputeBoW (src/sg-slam/src/Frame.cc:422-429) and KeyFrame::ComputeBoW (src/sg-slam/src/KeyFrame.cc:60-69):
 *     mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)          src/sg-slam/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1139-1206
 * The per-feature tree descent (:1231-1273, FORB::distance FORB.cpp:81-101) runs on the device; BowVector / FeatureVector assembly and scoring are host work.
 *
 * sgx_voc_load: loadFromTextFile (:1351-1438) for a path ending in ".txt", loadFromBinaryFile (:1467-1510) otherwise — System::System's rule (System.cc:69-73).
 * sgx_voc_create: the same tree from flat arrays: node 0 is the root, parent[i] < i for i >= 1, desc32 = nnodes x 32 bytes, is_leaf[i] gives node i the next word id. */
typedef struct sgx_voc sgx_voc;
int sgx_voc_load(const char *path, sgx_voc **out);
int sgx_voc_create(int k, int L, int scoring, int weighting, int nnodes, const int32_t *parent, const uint8_t *desc32, const double *weight, const uint8_t *is_leaf, sgx_voc **out);
int sgx_voc_info(const sgx_voc *v, int32_t *k, int32_t *L, int32_t *scoring, int32_t *weighting, int32_t *nnodes, int32_t *nwords);
void sgx_voc_destroy(sgx_voc *v);
/* transform(features, v, fv, levelsup) for one frame (host pointers, synchronous): desc = n x 32 bytes (mDescriptors rows).  bow_ids / bow_weights (capacity n) = the
 * BowVector in std::map order (word ids ascending; weights accumulated in feature order and normalised as the vocabulary's scoring demands), *nbow its size;
 * feat_node[i] = key of mFeatVec under which feature i sits (the node levelsup levels above its word), -1 when the word is stopped (weight <= 0, :1170) — exactly the
 * feat_node_* input of the sgx_match_search_by_bow* and sgx_match_search_for_triangulation entries; feat_word (optional) = word id per feature.
 * Divergence: when a descent reaches a leaf above level L - levelsup the reference leaves the node id unassigned (uninitialised variable); this build reports the leaf. */
int sgx_voc_transform(sgx_voc *v, int n, const uint8_t *desc, int levelsup, int32_t *bow_ids, double *bow_weights, int32_t *nbow, int32_t *feat_node, int32_t *feat_word);
/* the per-feature part for B frames resident on the device (frame b: d_n[b] descriptors at d_desc + b * desc_pitch; outputs with row pitch cap); asynchronous on `stream` */
int sgx_voc_transform_batch_dev(sgx_voc *v, const uint8_t *d_desc, size_t desc_pitch, const int32_t *d_n, int batch, int cap, int levelsup,
                                int32_t *d_word_id, double *d_weight, int32_t *d_feat_node, void *stream);
/* double TemplatedVocabulary::score(const BowVector&, const BowVector&) (:1210-1215) for the L1 scoring the ORB vocabulary files select (L1Scoring::score,
 * src/sg-slam/Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-67; callers KeyFrameDatabase.cc:133,249, LoopClosing.cc:134); SGX_ERR_UNSUPPORTED for the other scoring types */
int sgx_voc_score(const sgx_voc *v, int n1, const int32_t *ids1, const double *w1, int n2, const int32_t *ids2, const double *w2, double *score);

/* The colour conversion at the top of Tracking::GrabImageRGBD (src/sg-slam/src/Tracking.cc:214-227): cvtColor(CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) on
 * 8-bit images, OpenCV's fixed point (R*4899 + G*9617 + B*1868 + 8192) >> 14.  blue_first = !mbRGB.  Pitches in bytes, multiples of 4; pointers 4-byte aligned. */
int sgx_frame_gray_from_color_batch_dev(int batch, int width, int height, const uint8_t *d_src, int src_pitch, int channels, int blue_first,
                                        uint8_t *d_gray, int gray_pitch, void *stream);
/* ---- per-frame glue between the accelerated stages (device resident) -----------------------------
 * Frame::ComputeStereoFromRGBD (Frame.cc:893-914) fused with the u16 -> metres conversion
 * (Tracking.cc:229-230): uright[i] = x - bf/d, zdepth[i] = d, or -1 when depth is 0. */
int sgx_frame_stereo_from_rgbd_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n,
                                         const uint16_t *d_depth, int width, int height, float depth_map_factor,
                                         float bf, float *d_uright, float *d_zdepth, void *stream);
/* Frame::UnprojectStereo (Frame.cc:916-930) for every keypoint: xw = Rwc*x3Dc + Ow, has = depth>0 */
int sgx_frame_unproject_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_zdepth,
                                  const float *d_Tcw, const sgx_camera *cam, float *d_xw, uint8_t *d_has, void *stream);

/* Tracking's constant-velocity prediction (Tracking.cc:463-470, :914): pred = Tcur * inv(Tprev) * Tcur;
 * frames with valid[f]==0 (optional array) copy Tcur. */
int sgx_frame_motion_model_batch_dev(int batch, const float *d_Tcw_cur, const float *d_Tcw_prev, const uint8_t *d_valid,
                                     float *d_Tcw_pred, void *stream);

/* MapPoint::MapPoint(Pos, pMap, pFrame, idxF) (src/sg-slam/src/MapPoint.cc:45-67) for every keypoint with depth — the temporal points of
 * Tracking::UpdateLastFrame (Tracking.cc:840-904): normal, mfMin/MaxDistance, descriptor; written to slice `half` (0/1) of a
 * per-frame ring of 2*cap local-map records (the layout sgx_match_project_local_batch_dev consumes with mcap = 2*cap). */
int sgx_frame_make_map_points_batch_dev(int batch, int cap, int half, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_xw, const uint8_t *d_has,
                                        const uint8_t *d_desc, const float *d_Tcw, const float *scale_factors, int nlevels,
                                        float *d_m_xw, float *d_m_normal, float *d_m_min_dist, float *d_m_max_dist, uint8_t *d_m_desc, uint8_t *d_m_skip, void *stream);
/* mvpMapPoints after TrackWithMotionModel (+ SearchLocalPoints): merged[k] indexes the combined table xw_all = [last frame's points (cap) |
 * local-map ring (2*cap)]; motion-model outliers are dropped (Tracking.cc:941-956); a local-map match replaces an unobserved point.
 * cur_mp_obs[k] (optional out) = 0 for keypoints holding a visual-odometry point, -1 otherwise (input of the local-map matcher).
 * d_match_local / d_merged / d_xw_all may be NULL (first call of a frame only needs cur_mp_obs). */
int sgx_frame_merge_matches_batch_dev(int batch, int cap, const int32_t *d_n, const int32_t *d_match_last, const uint8_t *d_outlier_last, const int32_t *d_match_local,
                                      const float *d_xw_last, const float *d_m_xw, int32_t *d_merged, int32_t *d_cur_mp_obs, float *d_xw_all, void *stream);

/* ---- pose-only optimisation -------------------------------------------------------------------
 * Replaces `static int Optimizer::PoseOptimization(Frame *pFrame)` (src/sg-slam/include/Optimizer.h:50,
 * src/sg-slam/src/Optimizer.cc:239-451): g2o Levenberg-Marquardt over the frame pose with one
 * mono (mvuRight<0) or stereo reprojection edge per keypoint that holds a map point, Huber kernel,
 * 4 rounds x 10 iterations with chi2 inlier classification between rounds.  fp64 inside, fp32 at the
 * boundary exactly like the reference (Converter.cc:37-71).
 * Flattened Frame: keys_un = mvKeysUn, uright = mvuRight, inv_level_sigma2 = mvInvLevelSigma2;
 * mvpMapPoints[i] is given either per keypoint (has_mp[i], mp_xw[3*i..]) or through mp_index[i]
 * (row of mp_xw, -1 = NULL) so the matcher's output can be consumed without a gather.
 * In/out: Tcw (4x4 row-major float, pFrame->mTcw).  Out: outlier = mvbOutlier, return value in
 * n_inliers (nInitialCorrespondences - nBad; 0 when fewer than 3 correspondences). */
int sgx_pose_optimization_batch_dev(int batch, int cap, const sgx_keypoint *d_keys_un, const float *d_uright, const int32_t *d_n,
                                    const int32_t *d_mp_index, const uint8_t *d_has_mp, const float *d_mp_xw, int xw_pitch,
                                    const float *inv_level_sigma2, int nlevels, const sgx_camera *cam,
                                    float *d_Tcw, uint8_t *d_outlier, int32_t *d_n_inliers, void *stream);
int sgx_pose_optimization(int n, const sgx_keypoint *keys_un, const float *uright, const uint8_t *has_mp, const float *mp_xw,
                          const float *inv_level_sigma2, int nlevels, const sgx_camera *cam,
                          float *Tcw, uint8_t *outlier, int32_t *n_inliers);

/* ---- local bundle adjustment -------------------------------------------------------------------
 * Replaces `static void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool *pbStopFlag, Map *pMap)`
 * (src/sg-slam/include/Optimizer.h:52, src/sg-slam/src/Optimizer.cc:453-778) from the point where the local
 * graph is known: the caller flattens lLocalKeyFrames + lFixedCameras into `poses` (KeyFrame::mnId order, as g2o
 * orders vertices), lLocalMapPoints into `points`, and the observation edges in insertion order (outer loop over
 * points, Optimizer.cc:572-653).  optimize(5) with Huber -> chi2/depth classification -> optimize(10) without the
 * outliers -> per-edge erase flags (the (KeyFrame, MapPoint) pairs of vToErase, :709-757).  fp64 inside.
 * pose_fixed: 0 = local keyframe (optimised), 1 = fixed camera (lFixedCameras, never rewritten),
 *             2 = local keyframe with mnId==0 (fixed vertex, still rewritten through SE3Quat like :762-768).
 * edge_obs: (u, v, uR) per edge, uR < 0 => monocular edge.  stop_flag mirrors pbStopFlag (polled between trials). */
typedef struct sgx_ba_problem {
    int32_t n_poses, n_points, n_edges;
    float *poses;                 /* n_poses x 16 (Tcw row-major), in/out */
    const uint8_t *pose_fixed;    /* n_poses */
    float *points;                /* n_points x 3, in/out */
    const int32_t *edge_pose;     /* n_edges */
    const int32_t *edge_point;    /* n_edges */
    const float *edge_obs;        /* n_edges x 3 */
    const float *edge_info;       /* n_edges: mvInvLevelSigma2[octave] */
} sgx_ba_problem;
typedef struct sgx_ba_stats { int32_t iterations_first, iterations_second, free_poses, reserved; double chi2_first, chi2_second; } sgx_ba_stats;
int sgx_local_bundle_adjustment(const sgx_ba_problem *problem, const sgx_camera *cam, const volatile int32_t *stop_flag,
                                uint8_t *edge_erase, sgx_ba_stats *stats);
/* Replaces `static void Optimizer::BundleAdjustment(const vector<KeyFrame*> &vpKFs, const vector<MapPoint*> &vpMP, int nIterations, bool *pbStopFlag,
 * const unsigned long nLoopKF, const bool bRobust)` (src/sg-slam/include/Optimizer.h:44-46, src/sg-slam/src/Optimizer.cc:49-237; GlobalBundleAdjustemnt :40-47 is
 * the same call on all keyframes / map points of the Map): every non-bad keyframe is a pose (pose_fixed != 0 exactly for mnId == 0), every non-bad map point a
 * landmark, one edge per observation in a keyframe of the problem; ONE optimizer.optimize(nIterations) with Huber kernels sqrt(5.99) / sqrt(7.815) when bRobust, no
 * outlier classification.  Out: every pose (the fixed one included, :200-214) and every point that has at least one edge (:216-236; the others are
 * vbNotIncludedMP) — the caller stores them with SetPose / SetWorldPos (nLoopKF == 0) or in mTcwGBA / mPosGBA.  stats: iterations_first, chi2_first. */
int sgx_bundle_adjustment(const sgx_ba_problem *problem, const sgx_camera *cam, int n_iterations, const volatile int32_t *stop_flag, int robust, sgx_ba_stats *stats);

/* int Optimizer::OptimizeSim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches1, g2o::Sim3 &g2oS12, const float th2, const bool bFixScale)
 * (src/sg-slam/include/Optimizer.h:56-57, src/sg-slam/src/Optimizer.cc:1046-1257; caller LoopClosing::ComputeSim3, LoopClosing.cc:326).  The caller lists the n correspondences the
 * reference turns into edge pairs (vpMatches1[i] && both map points good && pMP2 observed in pKF2, :1107-1133) in index order: p1c / p2c = the two map points in their own
 * keyframe's camera frame (R1w*P3D1w + t1w, R2w*P3D2w + t2w), obs1 / obs2 = kpUn.pt, info1 / info2 = mvInvLevelSigma2[octave], K1 / K2 = (fx, fy, cx, cy).
 * S12 in/out = (qx, qy, qz, qw, tx, ty, tz, s) of g2o::Sim3; inlier[i] = 0 where the reference sets vpMatches1[idx] = NULL; iterations (optional) = LM iterations of the two
 * optimize() calls; *n_inliers = return value (0 with S12 untouched when fewer than 10 correspondences survive the first check).  Host pointers, synchronous. */
int sgx_optimize_sim3(int n, const float *p1c, const float *p2c, const float *obs1, const float *obs2, const float *info1, const float *info2,
                      const float *K1, const float *K2, double *S12, float th2, int fix_scale, uint8_t *inlier, int32_t *iterations, int32_t *n_inliers);

/* The optimisation of void Optimizer::OptimizeEssentialGraph(Map*, KeyFrame *pLoopKF, KeyFrame *pCurKF, const KeyFrameAndPose &NonCorrectedSim3, const KeyFrameAndPose &CorrectedSim3,
 * const map<KeyFrame*, set<KeyFrame*>> &LoopConnections, const bool &bFixScale) (src/sg-slam/include/Optimizer.h:49-52, src/sg-slam/src/Optimizer.cc:781-1042; caller
 * LoopClosing::CorrectLoop, LoopClosing.cc:567): solver->setUserLambdaInit(1e-16); optimize(20) on the pose graph the reference builds at :807-958.  The caller flattens that graph:
 * S_in[v] = the vertex estimate (vScw: the corrected Sim3 where CorrectedSim3 holds the keyframe, Sim3(Rcw, tcw, 1) otherwise) as (qx, qy, qz, qw, tx, ty, tz, s); fixed[v] != 0 for
 * pLoopKF; one row per g2o::EdgeSim3 in insertion order: e_i = vertex 0, e_j = vertex 1, e_meas = the measurement Sji.  S_out[v] = the optimised estimates (CorrectedSiw, :970-973);
 * stats (optional, 3 doubles): LM iterations, chi2 before, chi2 after.  EdgeSim3 has no analytic Jacobian in the vendored g2o: numeric differences, as the reference.  Host pointers. */
int sgx_optimize_essential_graph(int nv, const double *S_in, const uint8_t *fixed, int ne, const int32_t *e_i, const int32_t *e_j, const double *e_meas,
                                 int fix_scale, int iterations, double *S_out, double *stats);
/* The map-point correction that follows (Optimizer.cc:1004-1041): xw_out[i] = correctedSwr.map(Srw.map(xw[i])) with r = ref[i], the point's reference keyframe (or mnCorrectedReference);
 * Srw = vScw, corrected_Swr = vCorrectedSwc (nv x 8 each).  The pose recovery [R t/s; 0 1] (:975-985) is three divisions on the host. */
int sgx_correct_map_points(int n, const float *xw, const int32_t *ref, int nv, const double *Srw, const double *corrected_Swr, float *xw_out);

/* ---- 2-D detector + dynamic-feature mask --------------------------------------------------------
 * Replaces ORB_SLAM2::Detector2D (src/sg-slam/include/Detector2D.h:45-67, src/sg-slam/src/Detector2D.cc:16-89): the ncnn
 * forward pass of the shipped MobileNetV3-SSDLite graph (Thirdparty/ncnn_model/mobilenetv3_ssdlite_voc.param + .bin) on 300x300,
 * DetectionOutput, and detect()'s filtering into mvObjects2D / mvPotentialDynamicBorderForMapping / ...ForRmDynamicFeature.
 * sgx_det_create takes the .param TEXT and the .bin BYTES (the caller reads the two files Detector2D.cc:24-25 names);
 * pointwise convolutions run as fp32-MFMA GEMMs.  The Run()/isNewImageArrived()/ImageDetectFinished() thread handshake
 * (Detector2D.cc:122-149) becomes stream ordering: the caller launches detect on its own HIP stream / thread. */
#define SGX_DET_MAX 100
typedef struct sgx_detection { float label, score, xmin, ymin, xmax, ymax; } sgx_detection;        /* ncnn DetectionOutput row (normalised coords) */
typedef struct sgx_object2d { int32_t id; float prob, x, y, w, h; } sgx_object2d;                   /* Object2D: class id, prob, cv::Rect_<float> in pixels */
typedef struct sgx_det_result {
    int32_t n_raw; sgx_detection raw[SGX_DET_MAX];
    int32_t n_objects; sgx_object2d objects[SGX_DET_MAX];            /* mvObjects2D (non-person) */
    int32_t have_dynamic_for_mapping, have_dynamic_for_rm_feature;   /* mbHaveDynamicObjectFor{Mapping,RmDynamicFeature} */
    int32_t n_map_boxes; sgx_object2d map_boxes[SGX_DET_MAX];        /* mvPotentialDynamicBorderForMapping */
    int32_t n_rm_boxes; sgx_object2d rm_boxes[SGX_DET_MAX];          /* mvPotentialDynamicBorderForRmDynamicFeature (prob > 0.2) */
} sgx_det_result;
typedef struct sgx_det sgx_det;
int sgx_det_create(const char *param_text, const void *bin, size_t bin_bytes, int width, int height, int max_batch,
                   float detection_confidence_threshold, float dynamic_detection_confidence_threshold, sgx_det **out);
void sgx_det_destroy(sgx_det *h);
int sgx_det_info(const sgx_det *h, int32_t *num_priors, int32_t *num_class, int32_t *num_kernels, double *gmac);
/* Detector2D::detect(const cv::Mat &bgr) for `batch` host images (interleaved 3-channel u8, row pitch in bytes); synchronous */
int sgx_det_detect(sgx_det *h, const uint8_t *images, int pitch, int batch, sgx_det_result *results);
/* Detector2D::detect, device-resident and asynchronous on `stream`: forward + ncnn DetectionOutput (decode, per-class NMS, keep_top_k) + detect()'s
 * filtering, all on the GPU.  d_results: batch structs in device memory (same layout as the host entry fills).  d_boxes / d_nboxes / d_have_dynamic
 * (optional, may be NULL): the person rectangles (max_boxes x (x, y, w, h) per image), their count and mbHaveDynamicObjectForRmDynamicFeature, laid out
 * as sgx_dynamic_mask_batch_dev / sgx_frame_compact_keys_batch_dev take them — Detector2D.cc:53-88 feeding Frame.cc:482-500 without a host round trip.
 * Two documented differences from the reference, both where the reference's own result is not defined:
 *   - Frame.cc:482-491 copies the detector's flags into the Frame only when mvObjects2D (the NON-person list) is not empty; otherwise the Frame's
 *     mbHaveDynamicObjectForRmDynamicFeature (Frame.h:112, never initialised by the constructor) is read indeterminate at :493/:512/:565/:599.  d_have_dynamic[f]
 *     here is always Detector2D's own flag (a person with prob > 0.2 exists), i.e. the value the reference reads whenever it is defined.
 *   - ncnn's DetectionOutput orders candidates with an unstable quicksort (qsort_descent_inplace); rows with EXACTLY equal scores may come out in either order
 *     there.  Here (and in oracle/detector_oracle.py) ties keep class-major, then prior-index order (a stable sort); with distinct scores the outputs agree. */
int sgx_det_detect_batch_dev(sgx_det *h, const uint8_t *d_img, int pitch, int batch, sgx_det_result *d_results,
                             float *d_boxes, int32_t *d_nboxes, int max_boxes, int32_t *d_have_dynamic, void *stream);
/* device-resident batched forward only: leaves mbox_loc (num_priors*4) and softmax conf (num_priors*num_class) per image in HBM */
int sgx_det_forward_batch_dev(sgx_det *h, const uint8_t *d_img, int pitch, int batch, const float **d_loc, const float **d_conf, void *stream);
/* Matrix-product scheme of the detector's 1x1 convolutions (pointwise / expand / project / squeeze-excite), read by the NEXT sgx_det_create: 0 = exact fp32
 * (v_mfma_f32_32x32x2_f32, an ascending-k fmaf chain: bit-identical to the per-layer reference kernels), 1 = bf16x3 (each fp32 operand split exactly into three
 * bf16 terms, the six leading cross products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-accurate products in another summation order), -1 = the
 * default (1).  Selected per detector by the test tap sgx_det_debug_set_gemm (include/sgx_debug.h); sgx_det_gemm_mode reports what a detector was built with. */
int sgx_det_gemm_mode(const sgx_det *h);
/* one line of text per plan step i in [0, num_kernels of sgx_det_info): kind, ncnn layers, shapes, ' bf16x3' behind the steps that run on the bf16 matrix pipes (introspection for
 * profilers and the roofline accounting of bench.py); SGX_ERR_INVALID past the last step */
int sgx_det_plan_step(const sgx_det *h, int i, char *buf, int cap);
/* Frame::RmDynamicPointWithSemanticAndGeometry's keep/erase predicate (src/sg-slam/src/Frame.cc:556-597, :613-652): keep[i] = 1 when the
 * epipolar distance of (keypoint i, its LK-tracked previous position) under F (3x3 row-major fp64, cv::findFundamentalMat) is below
 * 0.2 px inside a person box / 1.0 px elsewhere (an all-zero F = "no fundamental matrix": the frame keeps every keypoint).  boxes: max_boxes x (x, y, w, h) per frame.  The caller applies the
 * "restore everything when fewer than 0.1*nFeatures survive" rule (:599-604) and compacts keypoints + descriptor rows. */
int sgx_dynamic_mask_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_prev_xy, const double *d_F,
                               const float *d_boxes, const int32_t *d_nboxes, int max_boxes, uint8_t *d_keep, void *stream);
/* The erase step that follows the mask (Frame.cc:556-604): keypoints with keep == 0 and their descriptor rows are removed, order preserved;
 * when have_dynamic[f] != 0 (a person box with prob > 0.2 exists, Detector2D.cc:80-84) and fewer than 0.1 * nfeatures keypoints survive, all
 * keypoints of the frame are restored (:599-604).  Out of place; d_have_dynamic may be NULL (no dynamic object in any frame). */
int sgx_frame_compact_keys_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const uint8_t *d_desc, const int32_t *d_n, const uint8_t *d_keep,
                                     const int32_t *d_have_dynamic, int nfeatures, sgx_keypoint *d_keys_out, uint8_t *d_desc_out, int32_t *d_n_out, void *stream);


/* ---- inputs of the dynamic-feature mask: optical flow + fundamental matrix ------------------------
 * What Frame::RmDynamicPointWithSemanticAndGeometry computes before its erase loop (src/sg-slam/src/Frame.cc:430-472):
 *   :445     cv::calcOpticalFlowPyrLK(imGray, imGrayPre, Curpoint, Prepoint, State, Err, Size(21,21), 3, TermCriteria(ITER|EPS, 30, 0.01))
 *            — every keypoint of the current frame tracked into the PREVIOUS frame (State / Err are ignored by the caller)
 *   :454-467 the pairs whose previous position lies outside the previous frame's person boxes (vPreFramePotentialDynamicBorder)
 *   :469-472 cv::findFundamentalMat(cur, prev, FM_RANSAC, 1.0, 0.99) on that selection when more than 20 pairs remain, else on all pairs
 * A sgx_flow handle owns the two image pyramids (current / previous) like the file-scope `imGrayPre` of Frame.cc:31,155-163: each
 * sgx_flow_lk_batch_dev call builds the pyramid of the new frames, tracks into the pyramid kept from the previous call, and swaps (the Scharr
 * derivatives are evaluated inside the tracker from the image itself).  OpenCV 3.4 semantics (lkpyramid.cpp, pyramids.cpp).
 * Divergence (stated, with the two undefined reference behaviours documented at sgx_det_detect and sgx_fundamental_ransac_batch_dev): the 2 x 2 gradient matrix and the
 * mismatch vector of every iteration are summed EXACTLY (int32 partial sums recombined in fp64, rounded once to fp32) — the order-free form of OpenCV's accumulation,
 * whose own float summation order depends on the SIMD path the build dispatches to (SSE2 / AVX2 / scalar `acctype`).  Against the oracle's OpenCV-order mode
 * (oracle/flow_oracle.c, acc_mode 0) tracked positions agree to 0.01 px on the test streams; mask decisions of knife-edge keypoints can therefore differ from a
 * given x86 build of the reference (bench.py reports the resulting trajectory difference as `ate_vs_oracle_chain`). */
typedef struct sgx_flow_config {
    int32_t width, height, max_batch;
    int32_t win_size;       /* 21 (only value supported) */
    int32_t max_level;      /* 3  (0..3) */
    int32_t max_count;      /* 30 */
    double epsilon;         /* 0.01 */
} sgx_flow_config;
typedef struct sgx_flow sgx_flow;
int sgx_flow_create(const sgx_flow_config *cfg, sgx_flow **out);
void sgx_flow_destroy(sgx_flow *h);
int sgx_flow_reset(sgx_flow *h);                 /* forget the previous frame (imGrayPre.data == NULL) */
int sgx_flow_levels(const sgx_flow *h);          /* pyramid levels actually used (buildOpticalFlowPyramid stops when a level is not larger than the window) */
/* d_gray: batch frames of height rows x pitch bytes (pitch and pointer multiples of 4).  When the handle holds a previous batch of the same size:
 * d_prev_xy[f*cap + i] = (x, y) of keypoint i of frame f in the previous frame (nextPts), d_status (optional) = State; *have_prev (host, optional) = 1.
 * First call after create / reset: only the pyramid is built, *have_prev = 0 (the reference skips the whole mask step, Frame.cc:155-163).
 * Asynchronous on `stream`. */
int sgx_flow_lk_batch_dev(sgx_flow *h, const uint8_t *d_gray, int pitch, int batch, const sgx_keypoint *d_keys, const int32_t *d_n, int cap,
                          float *d_prev_xy, uint8_t *d_status, int32_t *have_prev, void *stream);
/* cv::calcOpticalFlowPyrLK(gray_from, gray_to, pts, next_pts, status, ...) on one host image pair (n points, 2 floats each).  Synchronous; resets the streaming state. */
int sgx_flow_lk(sgx_flow *h, const uint8_t *gray_from, const uint8_t *gray_to, int stride, const float *pts, int n, float *next_pts, uint8_t *status);
/* Frame.cc:454-472 for `batch` frames: pair selection against the PREVIOUS frame's person boxes (d_pre_have_dynamic = bPreFrameHavePotentialDynamicObj,
 * d_pre_boxes / d_pre_nboxes = vPreFramePotentialDynamicBorder in the layout sgx_det_detect_batch_dev writes; all three may be NULL) and
 * cv::findFundamentalMat(FM_RANSAC, threshold, confidence): cv::RNG(-1) sampling, 7-point solver, symmetric epipolar error, adaptive iteration count,
 * no refit (fundam.cpp, ptsetreg.cpp).  d_F: 9 doubles per frame, row-major, x_prev^T F x_cur = 0 (what sgx_dynamic_mask_batch_dev takes);
 * d_ok[f] = 0 when OpenCV would return an empty Mat (fewer than 7 pairs, or no model) — F is then all zeros and the mask keeps every keypoint of
 * that frame (the reference indexes the empty Mat: undefined behaviour).  For 8..14 pairs OpenCV switches to LMeDSPointSetRegistrator::run and so does this entry
 * (same subsets from the same RNG, smallest median error, sigma-based inlier count, success = at least 7 inliers).
 * d_stats (optional): 4 ints per frame — RANSAC iterations run, iteration and root index of the returned model, its inlier count. */
int sgx_fundamental_ransac_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_prev_xy,
                                     const int32_t *d_pre_have_dynamic, const float *d_pre_boxes, const int32_t *d_pre_nboxes, int max_boxes,
                                     double threshold, double confidence, double *d_F, int32_t *d_ok, int32_t *d_stats, void *stream);
/* cv::findFundamentalMat(pts1, pts2, FM_RANSAC, threshold, confidence) on host points.  Synchronous. */
int sgx_find_fundamental_mat(const float *pts1, const float *pts2, int n, double threshold, double confidence, double *F, int32_t *ok, int32_t *stats);

/* ---- the pipelined per-frame host (C++ behind this C ABI: sg_slam_amd/csrc/sgx_tracker.cpp) ---------------------------------
 * S independent RGB-D streams tracked in lock-step on one GPU, one frame per stream per step, in the call order of the reference's tracking thread:
 * Tracking::GrabImageRGBD (src/sg-slam/src/Tracking.cc:206-251) -> Frame::Frame (src/sg-slam/src/Frame.cc:100-200: ExtractORB, detector hand-shake :170-176 / :478,
 * RmDynamicPointWithSemanticAndGeometry :430-610, ComputeStereoFromRGBD :893-914) -> Tracking::TrackWithMotionModel / TrackLocalMap (Tracking.cc:906-1013), in
 * visual-odometry form (Tracking::UpdateLastFrame :840-904 points; local map = the points of frames t-2, t-3).  Three HIP streams (detector | extraction + mask |
 * tracking) chained by events, triple-buffered frame state, no host synchronisation inside a step; an optional fourth stream uploads host frames.
 * It is the harness host of bench.py / example_track.cpp / the tests — keyframe decisions, relocalisation and the map stay with ORB_SLAM2::Tracking. */
typedef struct sgx_tracker_config {
    int32_t streams, width, height;
    int32_t nfeatures; float scale_factor; int32_t nlevels, ini_th_fast, min_th_fast;      /* ORBextractor.* of the settings file (TUM3.yaml: 1000, 1.2, 8, 20, 7) */
    sgx_camera cam; float depth_map_factor;                                                 /* Camera.*, DepthMapFactor (5000) */
    float th_projection;                                                                    /* SearchByProjection(cur, last, th): 15 (Tracking.cc:924) */
    int32_t local_map;                                                                      /* 1: TrackLocalMap stage (SearchLocalPoints + second PoseOptimization) */
    int32_t dynamic_mask;                                                                   /* 1: calcOpticalFlowPyrLK + findFundamentalMat + mask + erase (Frame.cc:430-610) */
    int32_t max_boxes;                                                                      /* person rectangles kept per frame (<= SGX_DET_MAX) */
    int32_t pipelined;                                                                      /* 1: own HIP streams; 0: everything on the caller's stream (tests) */
} sgx_tracker_config;
typedef struct sgx_tracker sgx_tracker;
int sgx_tracker_create(const sgx_tracker_config *cfg, sgx_det *detector /* NULL: no Detector2D::detect; not owned */, sgx_tracker **out);
void sgx_tracker_destroy(sgx_tracker *t);
int sgx_tracker_keypoint_capacity(const sgx_tracker *t);
int sgx_tracker_record_bytes(const sgx_tracker *t);                   /* 16 + cap * 28 + cap * 32 + 64: the per-frame record of BASELINE config 5 */
int sgx_tracker_set_initial_pose(sgx_tracker *t, const float *Tcw /* streams x 16, host */);
/* one frame of every stream from device memory: d_gray streams x height x gray_pitch u8, d_depth streams x height x width raw u16, d_bgr (optional, detector
 * input) streams x height x bgr_pitch interleaved 3-channel u8.  Asynchronous — a step only ENQUEUES work.  The inputs of a step may be rewritten
 * (a) by work enqueued on `caller_stream` after the THIRD following sgx_tracker_step_dev call (that call orders caller_stream behind the step's readers), or
 * (b) from the host after sgx_tracker_wait_inputs(t, steps_back) or sgx_tracker_sync returned. */
int sgx_tracker_step_dev(sgx_tracker *t, const uint8_t *d_gray, int gray_pitch, const uint16_t *d_depth, const uint8_t *d_bgr, int bgr_pitch, void *caller_stream);
/* host input (cv::imread's BGR image + the 16-bit depth map, rgbd_tum.cc:114-115): fill the pinned staging buffers of slot 0 / 1, then step; the tracker uploads
 * them on its own stream and converts to gray on the device (Tracking.cc:214-227; rgb_order = Camera.RGB).  The upload is asynchronous: before REFILLING a slot call
 * sgx_tracker_host_buffers(slot) again — it returns once the slot's pending upload has left the pinned buffers — or sgx_tracker_sync. */
int sgx_tracker_host_buffers(sgx_tracker *t, int slot, uint8_t **bgr, int *bgr_pitch, uint16_t **depth);
int sgx_tracker_step_host(sgx_tracker *t, int slot, int rgb_order);
/* blocks the host until the device has read the input images of the step issued `steps_back` (0..2) calls ago */
int sgx_tracker_wait_inputs(sgx_tracker *t, int steps_back);
int sgx_tracker_sync(sgx_tracker *t);
/* results of the frame tracked last (synchronises; any pointer may be NULL): Tcw streams x 16, keypoints after the mask, motion-model matches / inliers, local-map
 * matches / inliers of the second PoseOptimization, keypoints before the mask, findFundamentalMat success, its 4 statistics per stream */
int sgx_tracker_read(sgx_tracker *t, float *Tcw, int32_t *nkeys, int32_t *nmatches, int32_t *ninliers, int32_t *nmatches_local, int32_t *ninliers2, int32_t *nkeys_raw,
                     int32_t *f_ok, int32_t *f_stats);
int sgx_tracker_snapshot_pose_dev(sgx_tracker *t, float *d_out /* streams x 16 */);                               /* async copy on the tracking stream */
int sgx_tracker_snapshot_boxes_dev(sgx_tracker *t, int stream_index, float *d_boxes /* max_boxes x 4 */, int32_t *d_nboxes);   /* async copy on the detector stream */
/* (both snapshots copy on the tracker's own non-blocking streams: the destination buffers must have no pending writes on other streams — e.g. a zero-fill — when they are called) */
/* the frame records {n, cv::KeyPoint[cap], descriptors[cap][32], Tcw} of the frame tracked last, packed by one kernel into d_records (streams x record_bytes) on
 * `stream` after the frame's tracking event: what the RCCL gather of BASELINE config 5 sends */
int sgx_tracker_pack_records_dev(sgx_tracker *t, uint8_t *d_records, void *stream);
/* ---- the collective of BASELINE config 5 from the C++ host (round 6): gather of every rank's packed frame records to `root` over RCCL / xGMI ------------------------
 * One process per GPU.  Rank 0 calls sgx_dist_unique_id and hands the 128 bytes to the other ranks (the caller's launcher: MPI, a file, a socket); every rank then calls
 * sgx_dist_create on its own device.  sgx_dist_gather_records enqueues a grouped ncclSend / ncclRecv on `stream` — the stream sgx_tracker_pack_records_dev packed on — and
 * returns without synchronising: every rank sends bytes_per_rank = streams x record_bytes from d_send; root receives world such blocks side by side in d_recv (block r =
 * rank r's), the other ranks pass d_recv = NULL.  RCCL is loaded on first use (dlopen): a single-GPU program never needs it; SGX_ERR_UNSUPPORTED if it is absent.
 * Replaces nothing in the reference (it is single-GPU); the Python harness uses torch.distributed for the same pattern (sg_slam_amd/dist.py). */
typedef struct sgx_dist sgx_dist;
int sgx_dist_unique_id(void *id128 /* 128 bytes out */);
int sgx_dist_create(const void *id128, int world, int rank, sgx_dist **out);
void sgx_dist_destroy(sgx_dist *d);
int sgx_dist_world(const sgx_dist *d, int32_t *world, int32_t *rank);
int sgx_dist_gather_records(sgx_dist *d, const void *d_send, size_t bytes_per_rank, void *d_recv, int root, void *stream);
int sgx_tracker_frame_dev(sgx_tracker *t, const int32_t **d_n, const sgx_keypoint **d_keys, const uint8_t **d_desc, const float **d_Tcw, const float **d_xw, const uint8_t **d_has);
sgx_orb *sgx_tracker_extractor(sgx_tracker *t);

/* ---- per-kernel HIP-event timing (process-wide) ---------------------------------------------------
 * When enabled, every batched entry point records a hipEvent pair on the caller's stream around each
 * kernel launch.  sgx_profile_read sums the elapsed times per kernel class since the last reset
 * (ms[k], launches[k] for k < sgx_profile_num_classes(); it synchronises the device). */
int sgx_profile_enable(int on);
int sgx_profile_num_classes(void);
const char *sgx_profile_class_name(int k);
int sgx_profile_read(float *ms, int32_t *launches, int reset);

#ifdef __cplusplus
}
#endif
#endif
