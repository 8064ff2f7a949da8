// sgx_poseopt.cpp — host side of the PoseOptimization C-ABI (include/sgx.h).
// Reference behaviour: src/sg-slam/src/Optimizer.cc:239-451.
// fp64 solver arithmetic: multiply-adds may fuse here.  The reference does NOT fuse them — g2o and sg-slam are built plain -O3 (Thirdparty/g2o/CMakeLists.txt:57,
// src/sg-slam/CMakeLists.txt:11-12; only DBoW2 has -march=native) — a deliberate divergence inside the stated tolerance (1e-5 relative on the pose, identical
// outlier flags), see sgx_ba.cpp.  -DSGX_FP_CONTRACT_OFF keeps the reference's rounding.  The bit-exact integer / fp32 feature kernels keep -ffp-contract=off.
#ifndef SGX_FP_CONTRACT_OFF
#pragma clang fp contract(fast)
#endif
#include "sgx_poseopt_kernels.h"
#include "sgx_prof.h"
#include "sgx_stage.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <stdio.h>
#include <string.h>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

static SgxCam po_cam(const sgx_camera *c) { SgxCam k; k.fx = c->fx; k.fy = c->fy; k.cx = c->cx; k.cy = c->cy; k.bf = c->bf; k.minX = c->min_x; k.maxX = c->max_x; k.minY = c->min_y; k.maxY = c->max_y; return k; }

static thread_local int g_po_threads = 0;      // tuning / test tap: 0 = default (256), 64 or 256 = force
SGX_TAP int sgx_pose_opt_debug_set_threads(int t) { if (t != 0 && t != 64 && t != 256) return SGX_ERR_INVALID; g_po_threads = t; return SGX_OK; }

extern "C" int sgx_pose_optimization_batch_dev(int batch, int cap, const sgx_keypoint *d_keys_un, const float *d_uright, const int32_t *d_n,
                                                const int32_t *d_mp_index, const uint8_t *d_has_mp, const float *d_mp_xw, int xw_pitch,
                                                const float *inv_level_sigma2, int nlevels, const sgx_camera *cam,
                                                float *d_Tcw, uint8_t *d_outlier, int32_t *d_n_inliers, void *stream)
{
    if (batch < 1 || cap < 1 || cap > SGX_PO_CAP || !d_keys_un || !d_uright || !d_n || (!d_mp_index && !d_has_mp) || !d_mp_xw || xw_pitch < 1 ||
        !inv_level_sigma2 || nlevels < 1 || nlevels > 12 || !cam || !d_Tcw || !d_outlier || !d_n_inliers) return SGX_ERR_INVALID;
    SgxScales is2; memset(&is2, 0, sizeof is2);
    for (int i = 0; i < nlevels; i++) is2.s[i] = inv_level_sigma2[i];
    sgx_prof_begin(SGX_K_POSEOPT, (sgx_stream_t)stream);
    // threads per frame: four waves.  The one-wave variant (tap below) is kept for tuning: measured on MI355X it is 2x slower per launch at every
    // batch size (0.67 vs 0.35 ms at 64-256 frames, tools/bench_poseopt.py) because its 85 KB of LDS still limits a CU to one frame at a time.
    const int wide = g_po_threads ? (g_po_threads == 256) : 1;
    if (wide) { auto kfn = k_pose_opt<256>; SGX_LAUNCH(kfn, dim3(batch), dim3(256), (sgx_stream_t)stream, cap, (const uint8_t *)d_keys_un, d_uright, d_n,
                                                       d_mp_index, d_has_mp, d_mp_xw, xw_pitch, is2, po_cam(cam), d_Tcw, d_outlier, d_n_inliers); }
    else { auto kfn = k_pose_opt<64>; SGX_LAUNCH(kfn, dim3(batch), dim3(64), (sgx_stream_t)stream, cap, (const uint8_t *)d_keys_un, d_uright, d_n,
                                                 d_mp_index, d_has_mp, d_mp_xw, xw_pitch, is2, po_cam(cam), d_Tcw, d_outlier, d_n_inliers); }
    sgx_prof_end(SGX_K_POSEOPT, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

// Host pointers, one frame: the drop-in for `int Optimizer::PoseOptimization(Frame *pFrame)`.
extern "C" int sgx_pose_optimization(int n, const sgx_keypoint *keys_un, const float *uright, const uint8_t *has_mp, const float *mp_xw,
                                      const float *inv_level_sigma2, int nlevels, const sgx_camera *cam,
                                      float *Tcw, uint8_t *outlier, int32_t *n_inliers)
{
    if (n < 0 || n > SGX_PO_CAP || !Tcw || !outlier || !n_inliers || !cam || !inv_level_sigma2 || nlevels < 1) return SGX_ERR_INVALID;
    if (n > 0 && (!keys_un || !uright || !has_mp || !mp_xw)) return SGX_ERR_INVALID;      // a NULL source would leave the staging slot's previous contents in place
    const int cap = n > 0 ? n : 1;
    void *d[8] = {0};
    const size_t sz[8] = { (size_t)cap * 28, (size_t)cap * 4, 4, (size_t)cap, (size_t)cap * 12, 64, (size_t)cap, 4 };
    const void *src[8] = { keys_un, uright, &n, has_mp, mp_xw, Tcw, nullptr, nullptr };
    int rc = SGX_OK;
    for (int i = 0; i < 8 && rc == SGX_OK; i++) {
        if ((rc = sgx_stage().get(40 + i, sz[i], &d[i])) != SGX_OK) break;           // per-thread staging slots, kept from call to call (sgx_stage.h)
        if (src[i] && n > 0 && hipMemcpyAsync(d[i], src[i], i == 2 || i == 5 ? sz[i] : (size_t)n * (sz[i] / cap), hipMemcpyHostToDevice, 0) != hipSuccess) rc = SGX_ERR_DEVICE;
    }
    if (rc == SGX_OK && n == 0) { (void)hipMemcpyAsync(d[2], &n, 4, hipMemcpyHostToDevice, 0); (void)hipMemcpyAsync(d[5], Tcw, 64, hipMemcpyHostToDevice, 0); }
    if (rc == SGX_OK && hipStreamSynchronize(0) != hipSuccess) rc = SGX_ERR_DEVICE;
    if (rc == SGX_OK)
        rc = sgx_pose_optimization_batch_dev(1, cap, (const sgx_keypoint *)d[0], (const float *)d[1], (const int32_t *)d[2], nullptr, (const uint8_t *)d[3],
                                             (const float *)d[4], cap, inv_level_sigma2, nlevels, cam, (float *)d[5], (uint8_t *)d[6], (int32_t *)d[7], nullptr);
    if (rc == SGX_OK) {
        if (hipMemcpyAsync(Tcw, d[5], 64, hipMemcpyDeviceToHost, 0) != hipSuccess || hipMemcpyAsync(outlier, d[6], (size_t)n, hipMemcpyDeviceToHost, 0) != hipSuccess ||
            hipMemcpyAsync(n_inliers, d[7], 4, hipMemcpyDeviceToHost, 0) != hipSuccess || hipStreamSynchronize(0) != hipSuccess) rc = SGX_ERR_DEVICE;
    }
    return rc;
}

