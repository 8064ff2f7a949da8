// sgx_match.cpp — host side of the matcher / frame-glue C-ABI (include/sgx.h).
// Reference behaviour: src/sg-slam/src/ORBmatcher.cc:1332-1472, src/sg-slam/src/Frame.cc:893-932.
#include "sgx_match_kernels.h"
#include "sgx_prof.h"
#include "sgx_stage.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

static SgxCam to_cam(const sgx_camera *c) { SgxCam k; k.fx = c->fx; k.fy = c->fy; k.cx = c->cx; k.cy = c->cy; k.bf = c->bf; k.minX = c->min_x; k.maxX = c->max_x; k.minY = c->min_y; k.maxY = c->max_y; return k; }

extern "C" int sgx_match_project_frame_batch_dev(
    int batch, int cap,
    const sgx_keypoint *d_ckeys, const uint8_t *d_cdesc, const float *d_curight, const int32_t *d_cn, const float *d_cTcw,
    const sgx_keypoint *d_lkeys, const int32_t *d_ln, const uint8_t *d_l_has_mp, const uint8_t *d_l_outlier, const float *d_l_xw,
    const int32_t *d_l_obs, const uint8_t *d_l_mpdesc, const float *d_lTcw,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float th, int b_mono, int check_orientation,
    int32_t *d_cur_match, int32_t *d_nmatches, void *stream)
{
    if (batch < 1 || cap < 1 || cap > SGX_MATCH_CAP || !cam || !scale_factors || nlevels < 1 || nlevels > 12) return SGX_ERR_INVALID;
    if (!d_ckeys || !d_cdesc || !d_curight || !d_cn || !d_cTcw || !d_lkeys || !d_ln || !d_l_has_mp || !d_l_outlier || !d_l_xw ||
        !d_l_obs || !d_l_mpdesc || !d_lTcw || !d_cur_match || !d_nmatches) return SGX_ERR_INVALID;
    SgxScales sc; memset(&sc, 0, sizeof sc);
    for (int i = 0; i < nlevels; i++) sc.s[i] = scale_factors[i];
    sgx_prof_begin(SGX_K_MATCH, (sgx_stream_t)stream);
    static const int mthreads = sgx_getenv("SGX_TUNE_MATCH_THREADS") ? atoi(sgx_getenv("SGX_TUNE_MATCH_THREADS")) : SGX_MATCH_THREADS;   // env = tuning tap (64..1024)
    SGX_LAUNCH(k_match_project_frame, dim3(batch), dim3(mthreads), (sgx_stream_t)stream, cap,
               (const uint8_t *)d_ckeys, d_cdesc, d_curight, d_cn, d_cTcw, (const uint8_t *)d_lkeys, d_ln, d_l_has_mp, d_l_outlier,
               d_l_xw, d_l_obs, d_l_mpdesc, d_lTcw, to_cam(cam), sc, th, b_mono, check_orientation, d_cur_match, d_nmatches);
    sgx_prof_end(SGX_K_MATCH, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_frame_stereo_from_rgbd_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n,
                                                     const uint16_t *d_depth, int width, int height, float depth_map_factor,
                                                     float bf, float *d_uright, float *d_zdepth, void *stream)
{
    if (batch < 1 || cap < 1 || !d_keys || !d_n || !d_depth || !d_uright || !d_zdepth || !(depth_map_factor > 0)) return SGX_ERR_INVALID;
    const float inv = 1.0f / depth_map_factor;           // mDepthMapFactor = 1.0f/mDepthMapFactor, Tracking.cc:139-142
    sgx_prof_begin(SGX_K_STEREO, (sgx_stream_t)stream);
    SGX_LAUNCH(k_stereo_from_rgbd, dim3((cap + 255) / 256, batch), dim3(256), (sgx_stream_t)stream, cap, (const uint8_t *)d_keys, d_n,
               d_depth, width, height, inv, bf, d_uright, d_zdepth);
    sgx_prof_end(SGX_K_STEREO, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_frame_unproject_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_zdepth,
                                              const float *d_Tcw, const sgx_camera *cam, float *d_xw, uint8_t *d_has, void *stream)
{
    if (batch < 1 || cap < 1 || !d_keys || !d_n || !d_zdepth || !d_Tcw || !cam || !d_xw || !d_has) return SGX_ERR_INVALID;
    sgx_prof_begin(SGX_K_UNPROJECT, (sgx_stream_t)stream);
    SGX_LAUNCH(k_unproject, dim3((cap + 255) / 256, batch), dim3(256), (sgx_stream_t)stream, cap, (const uint8_t *)d_keys, d_n, d_zdepth,
               d_Tcw, to_cam(cam), d_xw, d_has);
    sgx_prof_end(SGX_K_UNPROJECT, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_match_project_local_batch_dev(
    int batch, int cap, const sgx_keypoint *d_ckeys, const uint8_t *d_cdesc, const float *d_curight, const int32_t *d_cn, const float *d_cTcw, const int32_t *d_cur_mp_obs,
    int mcap, const int32_t *d_mn, const float *d_m_xw, const float *d_m_normal, const float *d_m_min_dist, const float *d_m_max_dist, const uint8_t *d_m_desc,
    const int32_t *d_m_obs, const uint8_t *d_m_skip,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th, float nnratio, float viewing_cos_limit,
    int32_t *d_cur_match, int32_t *d_nmatches, uint8_t *d_in_view, void *stream)
{
    if (batch < 1 || cap < 1 || cap > SGX_MATCH_CAP || mcap < 1 || mcap > SGX_LOCAL_CAP || !cam || !scale_factors || nlevels < 1 || nlevels > 12) return SGX_ERR_INVALID;
    if (!d_ckeys || !d_cdesc || !d_curight || !d_cn || !d_cTcw || !d_mn || !d_m_xw || !d_m_normal || !d_m_min_dist || !d_m_max_dist || !d_m_desc || !d_m_obs ||
        !d_m_skip || !d_cur_match || !d_nmatches || !d_in_view) return SGX_ERR_INVALID;
    SgxScales sc; memset(&sc, 0, sizeof sc);
    for (int i = 0; i < nlevels; i++) sc.s[i] = scale_factors[i];
    sgx_prof_begin(SGX_K_MATCH_LOCAL, (sgx_stream_t)stream);
    SGX_LAUNCH(k_match_project_local, dim3(batch), dim3(SGX_MATCH_THREADS), (sgx_stream_t)stream, cap, (const uint8_t *)d_ckeys, d_cdesc, d_curight, d_cn, d_cTcw,
               d_cur_mp_obs, mcap, d_mn, d_m_xw, d_m_normal, d_m_min_dist, d_m_max_dist, d_m_desc, d_m_obs, d_m_skip, to_cam(cam), sc, nlevels, log_scale_factor,
               th, nnratio, viewing_cos_limit, d_cur_match, d_nmatches, d_in_view);
    sgx_prof_end(SGX_K_MATCH_LOCAL, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_frame_make_map_points_batch_dev(int batch, int cap, int half, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_xw, const uint8_t *d_has,
                                                    const uint8_t *d_desc, const float *d_Tcw, const float *scale_factors, int nlevels,
                                                    float *d_m_xw, float *d_m_normal, float *d_m_min_dist, float *d_m_max_dist, uint8_t *d_m_desc, uint8_t *d_m_skip, void *stream)
{
    if (batch < 1 || cap < 1 || half < 0 || half > 1 || !d_keys || !d_n || !d_xw || !d_has || !d_desc || !d_Tcw || !scale_factors || nlevels < 1 || nlevels > 12 ||
        !d_m_xw || !d_m_normal || !d_m_min_dist || !d_m_max_dist || !d_m_desc || !d_m_skip) return SGX_ERR_INVALID;
    SgxScales sc; memset(&sc, 0, sizeof sc);
    for (int i = 0; i < nlevels; i++) sc.s[i] = scale_factors[i];
    sgx_prof_begin(SGX_K_MAPGLUE, (sgx_stream_t)stream);
    SGX_LAUNCH(k_make_map_points, dim3((cap + 255) / 256, batch), dim3(256), (sgx_stream_t)stream, cap, half, (const uint8_t *)d_keys, d_n, d_xw, d_has, d_desc, d_Tcw, sc, nlevels,
               d_m_xw, d_m_normal, d_m_min_dist, d_m_max_dist, d_m_desc, d_m_skip);
    sgx_prof_end(SGX_K_MAPGLUE, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_frame_merge_matches_batch_dev(int batch, int cap, const int32_t *d_n, const int32_t *d_match_last, const uint8_t *d_outlier_last, const int32_t *d_match_local,
                                                  const float *d_xw_last, const float *d_m_xw, int32_t *d_merged, int32_t *d_cur_mp_obs, float *d_xw_all, void *stream)
{
    if (batch < 1 || cap < 1 || !d_n || !d_match_last || !d_outlier_last) return SGX_ERR_INVALID;
    sgx_prof_begin(SGX_K_MAPGLUE, (sgx_stream_t)stream);
    SGX_LAUNCH(k_merge_matches, dim3((cap + 255) / 256, batch), dim3(256), (sgx_stream_t)stream, cap, d_n, d_match_last, d_outlier_last, d_match_local, d_merged, d_cur_mp_obs);
    if (d_xw_all && d_xw_last && d_m_xw)
        SGX_LAUNCH(k_gather_xw, dim3((3 * cap + 255) / 256, batch), dim3(256), (sgx_stream_t)stream, cap, d_xw_last, d_m_xw, d_xw_all);
    sgx_prof_end(SGX_K_MAPGLUE, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_frame_motion_model_batch_dev(int batch, const float *d_Tcw_cur, const float *d_Tcw_prev, const uint8_t *d_valid, float *d_Tcw_pred, void *stream)
{
    if (batch < 1 || !d_Tcw_cur || !d_Tcw_prev || !d_Tcw_pred) return SGX_ERR_INVALID;
    sgx_prof_begin(SGX_K_MOTION, (sgx_stream_t)stream);
    SGX_LAUNCH(k_motion_model, dim3((batch + 63) / 64), dim3(64), (sgx_stream_t)stream, batch, d_Tcw_cur, d_Tcw_prev, d_valid, d_Tcw_pred);
    sgx_prof_end(SGX_K_MOTION, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

// Host-pointer, single pair: the drop-in for ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono).
namespace {
// device staging of the host-pointer entries: per-thread slots that persist from call to call (sgx_stage.h); slot ranges: frame matcher 0.., local matcher 20..
struct DevBuf {
    void *p = nullptr; int slot = 0;
    int put(const void *src, size_t n) { SgxStaged s; const int rc = s.put(slot, src, n); p = s.p; return rc; }
};
}

extern "C" int sgx_match_project_frame(
    int nc, const sgx_keypoint *ckeys, const uint8_t *cdesc, const float *curight, const float *cTcw,
    int nl, const sgx_keypoint *lkeys, const uint8_t *l_has_mp, const uint8_t *l_outlier, const float *l_xw,
    const int32_t *l_obs, const uint8_t *l_mpdesc, const float *lTcw,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float th, int b_mono, int check_orientation,
    int32_t *cur_match, int32_t *nmatches)
{
    if (nc < 0 || nl < 0 || nc > SGX_MATCH_CAP || nl > SGX_MATCH_CAP || !cur_match || !nmatches || !cTcw || !lTcw || !cam || !scale_factors || nlevels < 1) return SGX_ERR_INVALID;
    if (nc > 0 && (!ckeys || !cdesc || !curight)) return SGX_ERR_INVALID;
    if (nl > 0 && (!lkeys || !l_has_mp || !l_outlier || !l_xw || !l_obs || !l_mpdesc)) return SGX_ERR_INVALID;
    int cap = nc > nl ? nc : nl; if (cap < 1) cap = 1;
    DevBuf b[16]; for (int i = 0; i < 16; i++) b[i].slot = i;
    std::vector<uint8_t> pad;
    auto up = [&](int k, const void *src, size_t elem, int n) -> int {
        pad.assign((size_t)cap * elem, 0);
        if (n > 0 && src) memcpy(pad.data(), src, (size_t)n * elem);
        int rc = b[k].put(pad.data(), pad.size());
        if (rc == SGX_OK && hipStreamSynchronize(0) != hipSuccess) rc = SGX_ERR_DEVICE;
        return rc;
    };
    int rc;
#define UP(k, src, elem, n) if ((rc = up(k, src, elem, n)) != SGX_OK) return rc
    UP(0, ckeys, 28, nc); UP(1, cdesc, 32, nc); UP(2, curight, 4, nc); UP(3, lkeys, 28, nl); UP(4, l_has_mp, 1, nl); UP(5, l_outlier, 1, nl);
    UP(6, l_xw, 12, nl); UP(7, l_obs, 4, nl); UP(8, l_mpdesc, 32, nl);
#undef UP
    if ((rc = b[9].put(&nc, 4)) != SGX_OK) return rc;
    if ((rc = b[10].put(&nl, 4)) != SGX_OK) return rc;
    if ((rc = b[11].put(cTcw, 64)) != SGX_OK) return rc;
    if ((rc = b[12].put(lTcw, 64)) != SGX_OK) return rc;
    if ((rc = b[13].put(nullptr, (size_t)cap * 4)) != SGX_OK) return rc;
    if ((rc = b[14].put(nullptr, 4)) != SGX_OK) return rc;
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    rc = sgx_match_project_frame_batch_dev(1, cap, (const sgx_keypoint *)b[0].p, (const uint8_t *)b[1].p, (const float *)b[2].p, (const int32_t *)b[9].p,
                                           (const float *)b[11].p, (const sgx_keypoint *)b[3].p, (const int32_t *)b[10].p, (const uint8_t *)b[4].p,
                                           (const uint8_t *)b[5].p, (const float *)b[6].p, (const int32_t *)b[7].p, (const uint8_t *)b[8].p,
                                           (const float *)b[12].p, cam, scale_factors, nlevels, th, b_mono, check_orientation,
                                           (int32_t *)b[13].p, (int32_t *)b[14].p, nullptr);
    if (rc != SGX_OK) return rc;
    SGX_CHECK_HIP(hipMemcpyAsync(cur_match, b[13].p, (size_t)nc * 4, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipMemcpyAsync(nmatches, b[14].p, 4, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    return SGX_OK;
}

// Host pointers, one frame: the drop-in for Tracking::SearchLocalPoints' isInFrustum loop + ORBmatcher::SearchByProjection(Frame &F, vpMapPoints, th).
extern "C" int sgx_match_project_local(
    int nc, const sgx_keypoint *ckeys, const uint8_t *cdesc, const float *curight, const float *cTcw, const int32_t *cur_mp_obs,
    int nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const int32_t *m_obs, const uint8_t *m_skip,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th, float nnratio, float viewing_cos_limit,
    int32_t *cur_match, int32_t *nmatches, uint8_t *in_view)
{
    if (nc < 0 || nm < 0 || nc > SGX_MATCH_CAP || nm > SGX_LOCAL_CAP || !cur_match || !nmatches || !cTcw || !cam || !scale_factors) return SGX_ERR_INVALID;
    for (int i = 0; i < nc; i++) cur_match[i] = -1;
    if (in_view) memset(in_view, 0, (size_t)nm);
    *nmatches = 0;
    if (nc == 0 || nm == 0) return SGX_OK;                       // the reference's loops do not execute (ORBmatcher.cc:52-127)
    DevBuf b[18]; for (int i = 0; i < 18; i++) b[i].slot = 20 + i;
    int rc;
    std::vector<int32_t> no_obs;
    if (!cur_mp_obs) { no_obs.assign((size_t)nc, -1); cur_mp_obs = no_obs.data(); }
#define UP(k, src, bytes) if ((rc = b[k].put(src, (size_t)(bytes))) != SGX_OK) return rc
    UP(0, ckeys, (size_t)nc * 28); UP(1, cdesc, (size_t)nc * 32); UP(2, curight, (size_t)nc * 4); UP(3, &nc, 4); UP(4, cTcw, 64); UP(5, cur_mp_obs, (size_t)nc * 4);
    UP(6, &nm, 4); UP(7, m_xw, (size_t)nm * 12); UP(8, m_normal, (size_t)nm * 12); UP(9, m_min_dist, (size_t)nm * 4); UP(10, m_max_dist, (size_t)nm * 4);
    UP(11, m_desc, (size_t)nm * 32); UP(12, m_obs, (size_t)nm * 4); UP(13, m_skip, (size_t)nm);
    UP(14, nullptr, (size_t)nc * 4); UP(15, nullptr, 4); UP(16, nullptr, (size_t)nm);
#undef UP
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    rc = sgx_match_project_local_batch_dev(1, nc, (const sgx_keypoint *)b[0].p, (const uint8_t *)b[1].p, (const float *)b[2].p, (const int32_t *)b[3].p, (const float *)b[4].p,
                                           (const int32_t *)b[5].p, nm, (const int32_t *)b[6].p, (const float *)b[7].p, (const float *)b[8].p, (const float *)b[9].p,
                                           (const float *)b[10].p, (const uint8_t *)b[11].p, (const int32_t *)b[12].p, (const uint8_t *)b[13].p,
                                           cam, scale_factors, nlevels, log_scale_factor, th, nnratio, viewing_cos_limit,
                                           (int32_t *)b[14].p, (int32_t *)b[15].p, (uint8_t *)b[16].p, nullptr);
    if (rc != SGX_OK) return rc;
    SGX_CHECK_HIP(hipMemcpyAsync(cur_match, b[14].p, (size_t)nc * 4, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipMemcpyAsync(nmatches, b[15].p, 4, hipMemcpyDeviceToHost, 0));
    if (in_view) SGX_CHECK_HIP(hipMemcpyAsync(in_view, b[16].p, (size_t)nm, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    return SGX_OK;
}

extern "C" int sgx_frame_gray_from_color_batch_dev(int batch, int width, int height, const uint8_t *d_src, int src_pitch, int channels, int blue_first,
                                                   uint8_t *d_gray, int gray_pitch, void *stream)
{
    if (batch < 1 || width < 1 || height < 1 || !d_src || !d_gray || (channels != 3 && channels != 4) || src_pitch < width * channels || gray_pitch < width ||
        (src_pitch & 3) || (gray_pitch & 3) || ((uintptr_t)d_src & 3) || ((uintptr_t)d_gray & 3)) return SGX_ERR_INVALID;
    SGX_LAUNCH(k_gray_from_color, dim3((width + 255) / 256, height, batch), dim3(64), (sgx_stream_t)stream, width, height, d_src, src_pitch, channels, blue_first ? 1 : 0, d_gray, gray_pitch);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

SGX_TAP int sgx_debug_flow_affine_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_A, const float *d_shift, const float *d_boxes,
                                               int max_boxes, float *d_prev_xy, void *stream)
{
    if (batch < 1 || cap < 1 || !d_keys || !d_n || !d_A || !d_prev_xy) return SGX_ERR_INVALID;
    SGX_LAUNCH(k_flow_affine, dim3((cap + 255) / 256, batch), dim3(256), (sgx_stream_t)stream, cap, (const uint8_t *)d_keys, d_n, d_A, d_shift, d_boxes, max_boxes, d_prev_xy);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}
