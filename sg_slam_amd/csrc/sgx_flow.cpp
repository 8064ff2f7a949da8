// sgx_flow.cpp — host side of the mask-input C-ABI (include/sgx.h): pyramidal Lucas-Kanade flow current -> previous frame and the RANSAC
// fundamental matrix, i.e. what Frame::RmDynamicPointWithSemanticAndGeometry does before its erase loop (src/sg-slam/src/Frame.cc:430-472).
#include "sgx_flow_kernels.h"
#include "sgx_prof.h"
#include "sgx_stage.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

struct sgx_flow {
    sgx_flow_config cfg;
    SgxLkGeom g;
    uint8_t *img[2] = { nullptr, nullptr };     // two pyramid slots (current / previous), max_batch frames each
    int cur = 0;                                // slot the NEXT call writes
    int prev_batch = 0;                         // frames held by the other slot (0 = no previous frame: `if(imGrayPre.data)` false, Frame.cc:155)
};

extern "C" int sgx_flow_create(const sgx_flow_config *cfg, sgx_flow **out)
{
    if (!cfg || !out) return SGX_ERR_INVALID;
    *out = nullptr;
    if (cfg->width < 1 || cfg->height < 1 || cfg->max_batch < 1 || cfg->max_count < 0 || !(cfg->epsilon >= 0)) return SGX_ERR_INVALID;
    if (cfg->win_size != SGX_LK_WIN || cfg->max_level < 0 || cfg->max_level >= SGX_LK_MAXL) return SGX_ERR_UNSUPPORTED;      // Frame.cc:445 uses 21 and 3
    if (cfg->width <= 1 || cfg->height <= 1) return SGX_ERR_UNSUPPORTED;
    { const unsigned long long qw = (unsigned long long)((cfg->width + 3) >> 2); if (qw * qw * (unsigned long long)cfg->height >= (1ull << 32)) return SGX_ERR_UNSUPPORTED; }      // the pyramid kernels split their dword index with a 32-bit reciprocal (exact below this size: ~4 k x 4 k)
    sgx_flow *h = new (std::nothrow) sgx_flow();
    if (!h) return SGX_ERR_NOMEM;
    h->cfg = *cfg;
    SgxLkGeom &g = h->g;
    memset(&g, 0, sizeof g);
    // buildOpticalFlowPyramid (lkpyramid.cpp): halve until a level would not be larger than the window
    int w = cfg->width, hh = cfg->height; unsigned io = 0;
    for (int l = 0;; l++) {
        g.w[l] = w; g.h[l] = hh; g.pitch[l] = (w + 3) & ~3; g.ioff[l] = io;
        io += (unsigned)g.pitch[l] * (unsigned)hh;
        g.nl = l + 1;
        if (l == cfg->max_level) break;
        const int nw = (w + 1) / 2, nh = (hh + 1) / 2;
        if (nw <= cfg->win_size || nh <= cfg->win_size) break;
        w = nw; hh = nh;
    }
    g.img_stride = (io + 15u + 64u) & ~15u;       // + slack: the 4-dword row loads of k_lk_pyrdown may run a few bytes past a level's last row
    for (int s = 0; s < 2; s++)
        if (hipMalloc((void **)&h->img[s], (size_t)g.img_stride * cfg->max_batch) != hipSuccess) { sgx_flow_destroy(h); return SGX_ERR_NOMEM; }
    *out = h;
    return SGX_OK;
}

extern "C" void sgx_flow_destroy(sgx_flow *h)
{
    if (!h) return;
    for (int s = 0; s < 2; s++) if (h->img[s]) (void)hipFree(h->img[s]);
    delete h;
}

extern "C" int sgx_flow_reset(sgx_flow *h) { if (!h) return SGX_ERR_INVALID; h->prev_batch = 0; return SGX_OK; }
extern "C" int sgx_flow_levels(const sgx_flow *h) { return h ? h->g.nl : SGX_ERR_INVALID; }

// pyramid of `batch` frames into slot `slot`
static int build_pyramid(sgx_flow *h, int slot, const uint8_t *d_gray, int pitch, int batch, sgx_stream_t st)
{
    const SgxLkGeom &g = h->g;
    uint8_t *base = h->img[slot];
    sgx_prof_begin(SGX_K_LK_PYR, st);
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };      // ceil(2^32 / d): exact quotients for the kernels' index split (index < 2^21, d < 2^11)
    SGX_LAUNCH(k_lk_copy, dim3(((g.pitch[0] >> 2) * g.h[0] + 255) / 256, batch), dim3(256), st, d_gray, g.w[0], g.h[0], pitch, base, g.pitch[0], g.img_stride, magic(g.pitch[0] >> 2));
    for (int l = 1; l < g.nl; l++)
        SGX_LAUNCH(k_lk_pyrdown, dim3(((g.pitch[l] >> 2) * g.h[l] + 255) / 256, batch), dim3(256), st, (const uint8_t *)(base + g.ioff[l - 1]), g.w[l - 1], g.h[l - 1], g.pitch[l - 1],
                   g.img_stride, base + g.ioff[l], g.w[l], g.h[l], g.pitch[l], g.img_stride, magic(g.pitch[l] >> 2));
    sgx_prof_end(SGX_K_LK_PYR, st);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

static int track(sgx_flow *h, int cur_slot, int prev_slot, int batch, const sgx_keypoint *d_keys, const int32_t *d_n, int cap, float *d_prev_xy, uint8_t *d_status, sgx_stream_t st)
{
    SgxLkArgs A;
    A.cur_img = h->img[cur_slot]; A.prev_img = h->img[prev_slot];
    A.keys = (const uint8_t *)d_keys; A.n = d_n; A.cap = cap; A.prev_xy = d_prev_xy; A.status = d_status;
    A.max_count = h->cfg.max_count > 100 ? 100 : h->cfg.max_count;                        // SparsePyrLKOpticalFlowImpl::calc clamps the criteria
    double eps = h->cfg.epsilon > 10. ? 10. : h->cfg.epsilon;
    A.eps2 = eps * eps; A.min_eig = (float)1e-4;
    sgx_prof_begin(SGX_K_LK_TRACK, st);
    static const int kpw = sgx_getenv("SGX_LK_KPW") ? atoi(sgx_getenv("SGX_LK_KPW")) : 2;      // keypoints per wave: 2 (default) / 4 = k_lk_trackN, 1 = k_lk_track; same results
    A.batch = batch; A.kblocks = (cap + 4 * kpw - 1) / (4 * kpw);
#ifdef SGX_DEBUG_TAPS      // the one- and four-keypoints-per-wave mappings exist in the tap build only (same results; kept as the A/B arms of SGX_LK_KPW)
    if (kpw == 4) { auto kfn = k_lk_trackN<4>; SGX_LAUNCH(kfn, dim3((unsigned)A.kblocks * (unsigned)batch), dim3(256), st, h->g, A); }
    else if (kpw != 2) { A.kblocks = (cap + 3) / 4; SGX_LAUNCH(k_lk_track, dim3((unsigned)A.kblocks * (unsigned)batch), dim3(256), st, h->g, A); }
    else
#endif
    { auto kfn = k_lk_trackN<2>; SGX_LAUNCH(kfn, dim3((unsigned)A.kblocks * (unsigned)batch), dim3(256), st, h->g, A); }
    sgx_prof_end(SGX_K_LK_TRACK, st);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_flow_lk_batch_dev(sgx_flow *h, const uint8_t *d_gray, int pitch, int batch, const sgx_keypoint *d_keys, const int32_t *d_n, int cap,
                                     float *d_prev_xy, uint8_t *d_status, int32_t *have_prev, void *stream)
{
    if (!h || !d_gray || batch < 1 || batch > h->cfg.max_batch || pitch < h->cfg.width) return SGX_ERR_INVALID;
    if ((pitch & 3) || ((uintptr_t)d_gray & 3)) return SGX_ERR_INVALID;
    sgx_stream_t st = (sgx_stream_t)stream;
    const int cur = h->cur, prev = cur ^ 1;
    const bool tracked = h->prev_batch == batch;
    if (tracked && (!d_keys || !d_n || !d_prev_xy || cap < 1)) return SGX_ERR_INVALID;
    int rc = build_pyramid(h, cur, d_gray, pitch, batch, st);
    if (rc != SGX_OK) return rc;
    if (tracked) { rc = track(h, cur, prev, batch, d_keys, d_n, cap, d_prev_xy, d_status, st); if (rc != SGX_OK) return rc; }
    if (have_prev) *have_prev = tracked ? 1 : 0;
    h->cur = prev; h->prev_batch = batch;          // std::swap(imGrayPre, imGrayT), Frame.cc:158-162
    return SGX_OK;
}

// cv::calcOpticalFlowPyrLK(prevImg = gray_from, nextImg = gray_to, prevPts = pts, nextPts, status, ...) for one host image pair.  Synchronous.
extern "C" int sgx_flow_lk(sgx_flow *h, const uint8_t *gray_from, const uint8_t *gray_to, int stride, const float *pts, int n, float *next_pts, uint8_t *status)
{
    if (!h || !gray_from || !gray_to || stride < h->cfg.width || n < 0 || (n > 0 && (!pts || !next_pts))) return SGX_ERR_INVALID;
    if (n == 0) return SGX_OK;
    const int W = h->cfg.width, H = h->cfg.height, P = (W + 3) & ~3;
    std::vector<uint8_t> pack((size_t)P * H * 2, 0);
    for (int y = 0; y < H; y++) { memcpy(&pack[(size_t)y * P], gray_to + (size_t)y * stride, (size_t)W); memcpy(&pack[(size_t)(H + y) * P], gray_from + (size_t)y * stride, (size_t)W); }
    std::vector<sgx_keypoint> kp((size_t)n);
    for (int i = 0; i < n; i++) { memset(&kp[(size_t)i], 0, sizeof(sgx_keypoint)); kp[(size_t)i].x = pts[2 * i]; kp[(size_t)i].y = pts[2 * i + 1]; }
    SgxStaged dI, dK, dN, dO, dS;
    int rc;
    if ((rc = dI.put(0, pack.data(), pack.size())) != SGX_OK || (rc = dK.put(1, kp.data(), kp.size() * sizeof(sgx_keypoint))) != SGX_OK ||
        (rc = dN.put(2, &n, sizeof n)) != SGX_OK || (rc = dO.put(3, nullptr, (size_t)n * 8)) != SGX_OK || (rc = dS.put(4, nullptr, (size_t)n)) != SGX_OK) return rc;
    const int keep_cur = h->cur, keep_prev = h->prev_batch;
    if ((rc = build_pyramid(h, 1, (const uint8_t *)dI.p, P, 1, 0)) != SGX_OK) return rc;                        // to-image ("nextImg") -> slot 1
    if ((rc = build_pyramid(h, 0, (const uint8_t *)dI.p + (size_t)P * H, P, 1, 0)) != SGX_OK) return rc;          // from-image -> slot 0
    if ((rc = track(h, 0, 1, 1, (const sgx_keypoint *)dK.p, (const int32_t *)dN.p, n, (float *)dO.p, (uint8_t *)dS.p, 0)) != SGX_OK) return rc;
    SGX_CHECK_HIP(hipMemcpy(next_pts, dO.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (status) SGX_CHECK_HIP(hipMemcpy(status, dS.p, (size_t)n, hipMemcpyDeviceToHost));
    h->cur = keep_cur; h->prev_batch = 0; (void)keep_prev;       // the host call used both slots: the streaming state starts over
    return SGX_OK;
}

SGX_TAP int sgx_flow_debug_read_level(sgx_flow *h, int slot, int frame, int level, uint8_t *img /* w*h tight */)
{
    if (!h || slot < 0 || slot > 1 || frame < 0 || frame >= h->cfg.max_batch || level < 0 || level >= h->g.nl) return SGX_ERR_INVALID;
    const SgxLkGeom &g = h->g;
    SGX_CHECK_HIP(hipDeviceSynchronize());
    if (!img) return SGX_ERR_INVALID;
    {
        std::vector<uint8_t> t((size_t)g.pitch[level] * g.h[level]);
        SGX_CHECK_HIP(hipMemcpy(t.data(), h->img[slot] + (size_t)frame * g.img_stride + g.ioff[level], t.size(), hipMemcpyDeviceToHost));
        for (int y = 0; y < g.h[level]; y++) memcpy(img + (size_t)y * g.w[level], &t[(size_t)y * g.pitch[level]], (size_t)g.w[level]);
    }
    return SGX_OK;
}
SGX_TAP int sgx_flow_debug_level_size(const sgx_flow *h, int level, int32_t *w, int32_t *hh)
{
    if (!h || level < 0 || level >= h->g.nl || !w || !hh) return SGX_ERR_INVALID;
    *w = h->g.w[level]; *hh = h->g.h[level];
    return SGX_OK;
}

// Test tap (round 6): a co-runner that does nothing but bf16 matrix products (v_mfma_f32_32x32x16_bf16 back to back, two waves per SIMD, every CU), `launches` launches on `stream`,
// asynchronous.  Beside it, compiler-generated PACKED fp32 instructions of a co-resident wave returned wrong values in lanes 48-63 (profiles/r6_lk_priority_diagnosis.md): the library
// is built without them (-fno-slp-vectorize, Makefile), and tests/test_flow_gpu.py runs the LK tracker beside this kernel to keep it that way.
#if defined(SGX_DEBUG_TAPS) && !defined(SGX_EMU)
typedef float sgx_dbg_f16v __attribute__((ext_vector_type(16)));
typedef __bf16 sgx_dbg_bf8 __attribute__((ext_vector_type(8)));
SGX_KERNEL_OCC(256, 2) k_dbg_corun_bf16(float *o, int iters)
{
    const uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
    sgx_dbg_f16v acc = {0}; sgx_dbg_bf8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((x >> i) & 7); b[i] = (__bf16)(float)((y >> i) & 3); }
    for (int i = 0; i < iters; i++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    float s = 0.f; for (int i = 0; i < 16; i++) s += acc[i];
    o[(blockIdx.x & 1023) * 256 + threadIdx.x] = s;
}
#endif
SGX_TAP int sgx_debug_corun_bf16(int blocks, int iters, int launches, void *stream)
{
#if defined(SGX_DEBUG_TAPS) && !defined(SGX_EMU)
    static float *sink = nullptr;
    if (blocks < 1 || iters < 1 || launches < 1) return SGX_ERR_INVALID;
    if (!sink) SGX_CHECK_HIP(hipMalloc((void **)&sink, 1024 * 256 * sizeof(float)));
    for (int l = 0; l < launches; l++) SGX_LAUNCH(k_dbg_corun_bf16, dim3((unsigned)blocks), dim3(256), (sgx_stream_t)stream, sink, iters);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
#else
    (void)blocks; (void)iters; (void)launches; (void)stream;
    return SGX_ERR_UNSUPPORTED;
#endif
}

extern "C" int sgx_fundamental_ransac_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_prev_xy,
                                                const int32_t *d_pre_have_dynamic, const float *d_pre_boxes, const int32_t *d_pre_nboxes, int max_boxes,
                                                double threshold, double confidence, double *d_F, int32_t *d_ok, int32_t *d_stats, void *stream)
{
    if (batch < 1 || cap < 1 || !d_keys || !d_n || !d_prev_xy || !d_F || !d_ok || max_boxes < 0) return SGX_ERR_INVALID;
    if (cap > SGX_FM_MAXPTS) return SGX_ERR_UNSUPPORTED;
    SgxFmArgs A;
    A.keys = (const uint8_t *)d_keys; A.n = d_n; A.cap = cap; A.prev_xy = d_prev_xy;
    A.pre_have = d_pre_have_dynamic; A.pre_boxes = d_pre_boxes; A.pre_nboxes = d_pre_nboxes; A.max_boxes = max_boxes;
    if (threshold <= 0) threshold = 3;                                                     // findFundamentalMat (fundam.cpp)
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    A.threshold = threshold; A.confidence = confidence; A.max_iters = 1000;               // createRANSACPointSetRegistrator(cb, 7, thr, conf) -> maxIters 1000
    A.F = d_F; A.ok = d_ok; A.stats = d_stats;
    const size_t lds = (size_t)cap * 16 + 1024 + (((size_t)cap + 15) & ~(size_t)15);
    sgx_prof_begin(SGX_K_FM_RANSAC, (sgx_stream_t)stream);
    SGX_LAUNCH_DYN(k_fm_ransac, dim3(batch), dim3(256), lds, (sgx_stream_t)stream, A);
    sgx_prof_end(SGX_K_FM_RANSAC, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

// cv::findFundamentalMat(points1, points2, FM_RANSAC, threshold, confidence) for one host point set.  Synchronous.
extern "C" int sgx_find_fundamental_mat(const float *pts1, const float *pts2, int n, double threshold, double confidence, double *F, int32_t *ok, int32_t *stats)
{
    if (n < 0 || (n > 0 && (!pts1 || !pts2)) || !F || !ok) return SGX_ERR_INVALID;
    if (n > SGX_FM_MAXPTS) return SGX_ERR_UNSUPPORTED;
    const int cap = n > 0 ? n : 1;
    std::vector<sgx_keypoint> kp((size_t)cap);
    memset(kp.data(), 0, kp.size() * sizeof(sgx_keypoint));
    for (int i = 0; i < n; i++) { kp[(size_t)i].x = pts1[2 * i]; kp[(size_t)i].y = pts1[2 * i + 1]; }
    SgxStaged dK, dN, dP, dF, dO, dS;
    int rc;
    if ((rc = dK.put(0, kp.data(), kp.size() * sizeof(sgx_keypoint))) != SGX_OK || (rc = dN.put(1, &n, sizeof n)) != SGX_OK ||
        (rc = dP.put(2, n ? pts2 : nullptr, (size_t)cap * 8)) != SGX_OK || (rc = dF.put(3, nullptr, 72)) != SGX_OK || (rc = dO.put(4, nullptr, 4)) != SGX_OK ||
        (rc = dS.put(5, nullptr, 16)) != SGX_OK) return rc;
    rc = sgx_fundamental_ransac_batch_dev(1, cap, (const sgx_keypoint *)dK.p, (const int32_t *)dN.p, (const float *)dP.p, nullptr, nullptr, nullptr, 0, threshold, confidence,
                                          (double *)dF.p, (int32_t *)dO.p, (int32_t *)dS.p, nullptr);
    if (rc != SGX_OK) return rc;
    SGX_CHECK_HIP(hipMemcpy(F, dF.p, 72, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(ok, dO.p, 4, hipMemcpyDeviceToHost));
    if (stats) SGX_CHECK_HIP(hipMemcpy(stats, dS.p, 16, hipMemcpyDeviceToHost));
    return SGX_OK;
}
