// sgx_tracker.cpp — the pipelined per-frame host of the tracking hot path, in C++ behind the C ABI (include/sgx.h: sgx_tracker_*).
//
// One object = S independent RGB-D streams tracked in lock-step on one GPU, one frame per stream per step, every stage device-resident.
// It mirrors the call order of the reference's tracking thread for one frame,
//     Tracking::GrabImageRGBD (src/sg-slam/src/Tracking.cc:206-251)  ->  Frame::Frame (src/sg-slam/src/Frame.cc:100-200: ExtractORB, the detector
//     hand-shake, RmDynamicPointWithSemanticAndGeometry, ComputeStereoFromRGBD)  ->  Tracking::Track / TrackWithMotionModel / TrackLocalMap
//     (Tracking.cc:906-1013),
// in "visual odometry" form (the map points a frame is tracked against are the previous frame's keypoints unprojected with their measured depth,
// Tracking::UpdateLastFrame :840-904; the local map is the points of frames t-2 and t-3), with the stages spread over three HIP streams:
//     D  Detector2D::detect of frame t            (the reference runs it on its own thread, Detector2D::Run; Frame.cc:478 waits for it)
//     E  ORB extract -> LK flow -> RANSAC F -> [event: detector done] -> dynamic mask + erase -> stereo-from-RGBD              of frame t + 1
//     T  motion model -> SearchByProjection(cur,last) -> PoseOptimization -> SearchLocalPoints -> PoseOptimization -> unproject -> map points   of frame t
// Frame state is triple-buffered (frame t lives in slot t % 3); events order the streams; nothing synchronises with the host inside a step.
// An optional upload stream U takes host frames (pinned staging buffers the caller fills, as cv::imread would) to the device and converts
// BGR -> gray (Tracking.cc:214-227) ahead of the step (sgx_tracker_step_host).
// This is the harness / integration host for bench.py, example_track.cpp and the tests — it replaces sg_slam_amd/tracker.py's orchestration, not
// ORB_SLAM2::Tracking's state machine (keyframe decisions, relocalisation and the map stay with the caller).
#include "sgx_rt.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

#ifndef SGX_EMU
typedef hipEvent_t sgx_ev;
typedef hipStream_t sgx_st;
static int ev_create(sgx_ev *e) { return hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess ? 0 : 1; }
static void ev_destroy(sgx_ev e) { if (e) (void)hipEventDestroy(e); }
static void ev_record(sgx_ev e, sgx_st s) { (void)hipEventRecord(e, s); }
static void st_wait(sgx_st s, sgx_ev e) { (void)hipStreamWaitEvent(s, e, 0); }
static int st_create(sgx_st *s, int high_priority)
{
    if (high_priority) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi) == hipSuccess ? 0 : 1; }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess ? 0 : 1;
}
static void st_destroy(sgx_st s) { if (s) (void)hipStreamDestroy(s); }
static void st_sync(sgx_st s) { if (s) (void)hipStreamSynchronize(s); }
static void ev_sync(sgx_ev e) { if (e) (void)hipEventSynchronize(e); }
static void *host_alloc(size_t n) { void *p = nullptr; return hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
static void host_free(void *p) { (void)hipHostFree(p); }
#else
typedef int sgx_ev;
typedef void *sgx_st;
static int ev_create(sgx_ev *e) { *e = 0; return 0; }
static void ev_destroy(sgx_ev) {}
static void ev_record(sgx_ev, sgx_st) {}
static void st_wait(sgx_st, sgx_ev) {}
static int st_create(sgx_st *s, int) { *s = nullptr; return 0; }
static void st_destroy(sgx_st) {}
static void st_sync(sgx_st) {}
static void ev_sync(sgx_ev) {}
static void *host_alloc(size_t n) { return calloc(1, n ? n : 1); }
static void host_free(void *p) { free(p); }
#endif

#define TRK_CHECK(expr) do { const int _rc = (expr); if (_rc != SGX_OK) return _rc; } while (0)
#define TRK_HIP(expr) do { if ((expr) != hipSuccess) return SGX_ERR_DEVICE; } while (0)

// per-frame record of BASELINE config 5 / SURVEY.md §8(e): 16 B header (n, 3 pad) + cap x cv::KeyPoint (28 B) + cap x 32 B descriptors + 64 B pose
SGX_KERNEL(256) k_pack_frame_records(int cap, const int32_t *n, const uint32_t *keys, const uint32_t *desc, const uint32_t *Tcw, uint32_t *rec, int rec_words)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.x;
    uint32_t *r = rec + (size_t)f * rec_words;
    const uint32_t *k = keys + (size_t)f * cap * 7, *d = desc + (size_t)f * cap * 8, *T = Tcw + (size_t)f * 16;
    if (tid < 4) r[tid] = tid == 0 ? (uint32_t)n[f] : 0u;
    for (int i = tid; i < cap * 7; i += 256) r[4 + i] = k[i];
    for (int i = tid; i < cap * 8; i += 256) r[4 + cap * 7 + i] = d[i];
    if (tid < 16) r[4 + cap * 15 + tid] = T[tid];
    SGX_THREADS_END
}

struct sgx_tracker {
    sgx_tracker_config cfg;
    int S = 0, cap = 0, nlevels = 0, MB = 0;
    sgx_orb *ex = nullptr; sgx_flow *flow = nullptr; sgx_det *det = nullptr;      // the detector is the caller's
    float scale[16], inv_sigma2[16]; float log_scale = 0.f;
    std::vector<void *> dev, pinned;
    // triple-buffered frame state
    sgx_keypoint *keys[3]; uint8_t *desc[3]; int32_t *n[3]; float *uright[3], *zdepth[3], *xw[3]; uint8_t *has[3];
    float *Tcw[3];                          // cur, last, last-last (rotating)
    int32_t *match, *nmatch, *ninl, *zero_i4; uint8_t *outlier, *zero_u8, *vel_valid;
    float *lm_xw, *lm_normal, *lm_min, *lm_max; uint8_t *lm_desc, *lm_skip; int32_t *lm_obs, *lm_n;
    int32_t *match_local, *nmatch_local, *merged, *cur_mp_obs, *ninl2; uint8_t *in_view, *outlier2; float *xw_all;
    // mask stage
    sgx_keypoint *rkeys; uint8_t *rdesc, *keep, *lk_status; int32_t *rn; float *prev_xy; double *F; int32_t *f_ok, *f_stats;
    float *pre_boxes; int32_t *pre_nboxes, *pre_have;
    float *no_boxes; int32_t *no_nboxes, *no_have;
    // detector results, double-buffered
    sgx_det_result *det_res[2]; float *det_boxes[2]; int32_t *det_nb[2], *det_have[2];
    // host-input staging (double-buffered): pinned host + device copies
    uint8_t *h_bgr[2] = { nullptr, nullptr }; uint16_t *h_depth[2] = { nullptr, nullptr };
    uint8_t *d_bgr[2] = { nullptr, nullptr }, *d_gray[2] = { nullptr, nullptr }; uint16_t *d_depth[2] = { nullptr, nullptr };
    int bgr_pitch = 0;
    sgx_st sE = nullptr, sT = nullptr, sD = nullptr, sU = nullptr;
    sgx_ev ev_extract[3] = {}, ev_track[3] = {}, ev_pack[3] = {}, ev_det[2] = {}, ev_up[2] = {}, ev_in = {};
    // ev_consumed[c]: every reader of the INPUT images of the step that used frame slot c (extraction stream: ORB, LK pyramid, stereo-from-RGBD; detector stream: the forward) is done.
    // ev_up[slot]: the H2D copies out of pinned staging slot `slot` are done (the caller may refill it).  ev_step: end of a non-pipelined step on the caller's stream.
    // ev_det_read[c]: the detector stream's reader of the same step (the forward reads d_bgr).  Kept apart from ev_consumed so that the extraction stream never waits for the
    // detector only to publish "inputs consumed" (with dynamic_mask == 0 that serialised ORB / LK of frame i + 1 behind the forward of frame i).
    sgx_ev ev_consumed[3] = {}, ev_det_read[3] = {}, ev_step = {};
    bool consumed_valid[3] = { false, false, false }, det_read_valid[3] = { false, false, false }, up_pending[2] = { false, false };
    sgx_st last_stream = nullptr;           // non-pipelined mode: the caller stream of the last step (snapshots and the record pack are ordered behind it)
    bool pack_pending[3] = { false, false, false };
    int frame_idx = 0, cur = 0;
    bool pipelined = false, shared_T = false;      // shared_T: sT aliases another stream (not destroyed separately)

    template <class Tp> int alloc(Tp **p, size_t count)
    {
        void *q = nullptr;
        if (hipMalloc(&q, (count ? count : 1) * sizeof(Tp)) != hipSuccess) return SGX_ERR_NOMEM;
        if (hipMemset(q, 0, (count ? count : 1) * sizeof(Tp)) != hipSuccess) return SGX_ERR_DEVICE;
        dev.push_back(q); *p = (Tp *)q; return SGX_OK;
    }
};

extern "C" void sgx_tracker_destroy(sgx_tracker *t)
{
    if (!t) return;
    st_sync(t->sE); st_sync(t->sT); st_sync(t->sD); st_sync(t->sU);
    (void)hipDeviceSynchronize();
    if (t->ex) sgx_orb_destroy(t->ex);
    if (t->flow) sgx_flow_destroy(t->flow);
    for (void *p : t->dev) (void)hipFree(p);
    for (void *p : t->pinned) host_free(p);
    for (int i = 0; i < 3; i++) { ev_destroy(t->ev_extract[i]); ev_destroy(t->ev_track[i]); ev_destroy(t->ev_pack[i]); ev_destroy(t->ev_consumed[i]); ev_destroy(t->ev_det_read[i]); }
    ev_destroy(t->ev_step);
    for (int i = 0; i < 2; i++) { ev_destroy(t->ev_det[i]); ev_destroy(t->ev_up[i]); }
    ev_destroy(t->ev_in);
    st_destroy(t->sE); if (!t->shared_T) st_destroy(t->sT); st_destroy(t->sD); st_destroy(t->sU);
    delete t;
}

extern "C" int sgx_tracker_create(const sgx_tracker_config *cfg, sgx_det *detector, sgx_tracker **out)
{
    if (!cfg || !out || cfg->streams < 1 || cfg->width < 32 || cfg->height < 32 || cfg->nfeatures < 1 || cfg->max_boxes < 1 || cfg->max_boxes > SGX_DET_MAX) return SGX_ERR_INVALID;
    sgx_tracker *t = new sgx_tracker();
    t->cfg = *cfg; t->S = cfg->streams; t->MB = cfg->max_boxes; t->det = detector;
    sgx_orb_config oc; memset(&oc, 0, sizeof oc);
    oc.nfeatures = cfg->nfeatures; oc.scale_factor = cfg->scale_factor; oc.nlevels = cfg->nlevels; oc.ini_th_fast = cfg->ini_th_fast; oc.min_th_fast = cfg->min_th_fast;
    oc.width = cfg->width; oc.height = cfg->height; oc.max_batch = cfg->streams;
#define FAIL(rc_) do { const int _r = (rc_); sgx_tracker_destroy(t); return _r; } while (0)
    int rc = sgx_orb_create(&oc, &t->ex); if (rc != SGX_OK) FAIL(rc);
    t->cap = sgx_orb_keypoint_capacity(t->ex); t->nlevels = cfg->nlevels;
    if (t->nlevels > 16) FAIL(SGX_ERR_INVALID);
    {
        rc = sgx_orb_get_tables(t->ex, t->scale, nullptr, nullptr, t->inv_sigma2, nullptr); if (rc != SGX_OK) FAIL(rc);
        t->log_scale = logf(t->scale[1]);                         // Frame::mfLogScaleFactor = log(mfScaleFactor) (Frame.cc:139)
    }
    const int S = t->S, cap = t->cap, MB = t->MB;
    for (int i = 0; i < 3; i++) {
        if (t->alloc(&t->keys[i], (size_t)S * cap) || t->alloc(&t->desc[i], (size_t)S * cap * 32) || t->alloc(&t->n[i], S) || t->alloc(&t->uright[i], (size_t)S * cap) ||
            t->alloc(&t->zdepth[i], (size_t)S * cap) || t->alloc(&t->xw[i], (size_t)S * cap * 3) || t->alloc(&t->has[i], (size_t)S * cap) || t->alloc(&t->Tcw[i], (size_t)S * 16)) FAIL(SGX_ERR_NOMEM);
    }
    if (t->alloc(&t->match, (size_t)S * cap) || t->alloc(&t->nmatch, S) || t->alloc(&t->ninl, S) || t->alloc(&t->zero_i4, (size_t)S * cap) || t->alloc(&t->outlier, (size_t)S * cap) ||
        t->alloc(&t->zero_u8, (size_t)S * cap) || t->alloc(&t->vel_valid, S)) FAIL(SGX_ERR_NOMEM);
    if (t->alloc(&t->lm_xw, (size_t)S * 2 * cap * 3) || t->alloc(&t->lm_normal, (size_t)S * 2 * cap * 3) || t->alloc(&t->lm_min, (size_t)S * 2 * cap) || t->alloc(&t->lm_max, (size_t)S * 2 * cap) ||
        t->alloc(&t->lm_desc, (size_t)S * 2 * cap * 32) || t->alloc(&t->lm_skip, (size_t)S * 2 * cap) || t->alloc(&t->lm_obs, (size_t)S * 2 * cap) || t->alloc(&t->lm_n, S) ||
        t->alloc(&t->match_local, (size_t)S * cap) || t->alloc(&t->nmatch_local, S) || t->alloc(&t->merged, (size_t)S * cap) || t->alloc(&t->cur_mp_obs, (size_t)S * cap) ||
        t->alloc(&t->ninl2, S) || t->alloc(&t->in_view, (size_t)S * 2 * cap) || t->alloc(&t->outlier2, (size_t)S * cap) || t->alloc(&t->xw_all, (size_t)S * 3 * cap * 3)) FAIL(SGX_ERR_NOMEM);
    {   // the local-map ring starts empty: every record skipped, Observations() = 1, 2 * cap records per stream
        std::vector<uint8_t> ones((size_t)S * 2 * cap, 1); std::vector<int32_t> one((size_t)S * 2 * cap, 1), full(S, 2 * cap);
        if (hipMemcpy(t->lm_skip, ones.data(), ones.size(), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(t->lm_obs, one.data(), one.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(t->lm_n, full.data(), full.size() * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
    }
    if (t->alloc(&t->rkeys, (size_t)S * cap) || t->alloc(&t->rdesc, (size_t)S * cap * 32) || t->alloc(&t->keep, (size_t)S * cap) || t->alloc(&t->lk_status, (size_t)S * cap) ||
        t->alloc(&t->rn, S) || t->alloc(&t->prev_xy, (size_t)S * cap * 2) || t->alloc(&t->F, (size_t)S * 9) || t->alloc(&t->f_ok, S) || t->alloc(&t->f_stats, (size_t)S * 4) ||
        t->alloc(&t->pre_boxes, (size_t)S * MB * 4) || t->alloc(&t->pre_nboxes, S) || t->alloc(&t->pre_have, S) || t->alloc(&t->no_boxes, (size_t)S * MB * 4) ||
        t->alloc(&t->no_nboxes, S) || t->alloc(&t->no_have, S)) FAIL(SGX_ERR_NOMEM);
    for (int i = 0; i < 2; i++)
        if (t->alloc(&t->det_res[i], S) || t->alloc(&t->det_boxes[i], (size_t)S * MB * 4) || t->alloc(&t->det_nb[i], S) || t->alloc(&t->det_have[i], S)) FAIL(SGX_ERR_NOMEM);
    if (cfg->dynamic_mask) {
        sgx_flow_config fc; memset(&fc, 0, sizeof fc);
        fc.width = cfg->width; fc.height = cfg->height; fc.max_batch = S; fc.win_size = 21; fc.max_level = 3; fc.max_count = 30; fc.epsilon = 0.01;      // Frame.cc:445
        rc = sgx_flow_create(&fc, &t->flow); if (rc != SGX_OK) FAIL(rc);
    }
    t->pipelined = cfg->pipelined != 0;
    if (t->pipelined) {
        // stream priorities (bit 0 extraction, 1 tracking, 2 detector; SGX_TRK_PRIO is the tuning tap).  Default: all equal.  Rounds 2-5 gave the tracking stream the high
        // priority (latency of a single camera); no throughput difference was ever measured.  Round 6 first blamed mixed priorities for a rare LK difference between co-running
        // trackers; they only changed how often the LK kernel met the detector's bf16 blocks on a CU — the cause was compiler-generated packed fp32 (profiles/r6_lk_priority_diagnosis.md).
        static const int prio = sgx_getenv("SGX_TRK_PRIO") ? atoi(sgx_getenv("SGX_TRK_PRIO")) : 0;
        // SGX_TRK_SHARE (tuning tap): 1 = tracking on the DETECTOR's stream (det(t), then track(t) behind it), 2 = tracking on the EXTRACTION stream (the round-1 serial order)
        static const int share = sgx_getenv("SGX_TRK_SHARE") ? atoi(sgx_getenv("SGX_TRK_SHARE")) : 0;
        if (st_create(&t->sE, prio & 1) || st_create(&t->sD, (prio >> 2) & 1)) FAIL(SGX_ERR_DEVICE);
        if (share == 1) { t->sT = t->sD; t->shared_T = true; } else if (share == 2) { t->sT = t->sE; t->shared_T = true; }
        else if (st_create(&t->sT, (prio >> 1) & 1)) FAIL(SGX_ERR_DEVICE);
        for (int i = 0; i < 3; i++) if (ev_create(&t->ev_extract[i]) || ev_create(&t->ev_track[i]) || ev_create(&t->ev_pack[i]) || ev_create(&t->ev_consumed[i]) || ev_create(&t->ev_det_read[i])) FAIL(SGX_ERR_DEVICE);
        for (int i = 0; i < 2; i++) if (ev_create(&t->ev_det[i])) FAIL(SGX_ERR_DEVICE);
        if (ev_create(&t->ev_in)) FAIL(SGX_ERR_DEVICE);
    } else if (ev_create(&t->ev_step)) FAIL(SGX_ERR_DEVICE);
    for (int i = 0; i < 2; i++) if (ev_create(&t->ev_up[i])) FAIL(SGX_ERR_DEVICE);      // both modes: the refill contract of sgx_tracker_host_buffers
#undef FAIL
    (void)hipDeviceSynchronize();           // the zero-fills above ran on the null stream, which the tracker's non-blocking streams do not wait for
    *out = t;
    return SGX_OK;
}

extern "C" int sgx_tracker_keypoint_capacity(const sgx_tracker *t) { return t ? t->cap : 0; }
extern "C" int sgx_tracker_record_bytes(const sgx_tracker *t) { return t ? 16 + t->cap * 28 + t->cap * 32 + 64 : 0; }

extern "C" int sgx_tracker_set_initial_pose(sgx_tracker *t, const float *Tcw)
{
    if (!t || !Tcw) return SGX_ERR_INVALID;
    for (int i = 0; i < 3; i++) TRK_HIP(hipMemcpy(t->Tcw[i], Tcw, (size_t)t->S * 16 * 4, hipMemcpyHostToDevice));
    return SGX_OK;
}

// One frame of every stream.  d_gray: S x H x gray_pitch u8; d_depth: S x H x W u16 (raw, Tracking.cc:229-230 divides by DepthMapFactor);
// d_bgr (optional, with a detector): S x H x bgr_pitch interleaved 3-channel u8 — what Detector2D::detect sees.  Asynchronous: the inputs are read on the
// tracker's streams after everything already enqueued on `caller_stream`.  Issuing a step only ENQUEUES work, so "three more steps were issued" says nothing about
// the device having read the inputs: the inputs of a step may be rewritten (a) by work enqueued on `caller_stream` after the third following
// sgx_tracker_step_dev call — that call makes `caller_stream` wait for the readers' event —, or (b) from the host after sgx_tracker_wait_inputs / sgx_tracker_sync returned.
extern "C" int sgx_tracker_step_dev(sgx_tracker *t, const uint8_t *d_gray, int gray_pitch, const uint16_t *d_depth, const uint8_t *d_bgr, int bgr_pitch, void *caller_stream)
{
    if (!t || !d_gray || !d_depth || gray_pitch < t->cfg.width) return SGX_ERR_INVALID;
    const int S = t->S, cap = t->cap, MB = t->MB, i = t->frame_idx, c = i % 3, l = (i + 2) % 3;
    const sgx_tracker_config &cf = t->cfg;
    float *Tc = t->Tcw[0], *Tl = t->Tcw[1], *Tll = t->Tcw[2];
    sgx_st sE = t->pipelined ? t->sE : (sgx_st)caller_stream, sT = t->pipelined ? t->sT : (sgx_st)caller_stream, sD = t->pipelined ? t->sD : (sgx_st)caller_stream;
    if (t->pipelined) {
        ev_record(t->ev_in, (sgx_st)caller_stream);                       // the frames may have been produced on the caller's stream
        st_wait(sE, t->ev_in); st_wait(sD, t->ev_in);
        if (i == 0) st_wait(sT, t->ev_in);
        if (t->consumed_valid[c]) st_wait((sgx_st)caller_stream, t->ev_consumed[c]);      // the inputs of step i - 3: later work on the caller's stream may overwrite them
        if (t->det_read_valid[c]) { st_wait((sgx_st)caller_stream, t->ev_det_read[c]); t->det_read_valid[c] = false; }
        if (i >= 2) st_wait(sE, t->ev_track[(i - 2) % 3]);                // slot c was "last" of step i - 2 + 1: its readers must be done
        if (t->pack_pending[c]) { st_wait(sE, t->ev_pack[c]); t->pack_pending[c] = false; }      // ... and its record must have been packed for the gather
    }
    // ---- D: Detector2D::detect of this frame (Frame.cc:170-176 hands the image to the detector thread)
    const bool with_det = t->det != nullptr && d_bgr != nullptr;
    const int b = i & 1;
    if (with_det) {
        if (t->pipelined && i >= 2) st_wait(sD, t->ev_extract[(i - 2) % 3]);      // result set b was last read by the mask / pre-box copy of step i - 2 (extraction stream)
        TRK_CHECK(sgx_det_detect_batch_dev(t->det, d_bgr, bgr_pitch, S, t->det_res[b], t->det_boxes[b], t->det_nb[b], MB, t->det_have[b], sD));
        if (t->pipelined) { ev_record(t->ev_det[b], sD); ev_record(t->ev_det_read[c], sD); t->det_read_valid[c] = true; }
    }
    const float *boxes = with_det ? t->det_boxes[b] : t->no_boxes; const int32_t *nboxes = with_det ? t->det_nb[b] : t->no_nboxes, *have = with_det ? t->det_have[b] : t->no_have;
    // ---- E: Frame::ExtractORB, then Frame::RmDynamicPointWithSemanticAndGeometry, then ComputeStereoFromRGBD
    if (cf.dynamic_mask) {
        if (i == 0) {
            TRK_CHECK(sgx_orb_extract_batch_dev(t->ex, d_gray, gray_pitch, S, t->keys[c], t->desc[c], t->n[c], cap, sE));
            TRK_CHECK(sgx_flow_lk_batch_dev(t->flow, d_gray, gray_pitch, S, nullptr, nullptr, cap, nullptr, nullptr, nullptr, sE));      // first frame: pyramid only (imGrayPre empty, Frame.cc:155-163)
        } else {
            TRK_CHECK(sgx_orb_extract_batch_dev(t->ex, d_gray, gray_pitch, S, t->rkeys, t->rdesc, t->rn, cap, sE));
            TRK_CHECK(sgx_flow_lk_batch_dev(t->flow, d_gray, gray_pitch, S, t->rkeys, t->rn, cap, t->prev_xy, t->lk_status, nullptr, sE));
            TRK_CHECK(sgx_fundamental_ransac_batch_dev(S, cap, t->rkeys, t->rn, t->prev_xy, t->pre_have, t->pre_boxes, t->pre_nboxes, MB, 1.0, 0.99, t->F, t->f_ok, t->f_stats, sE));
            if (with_det && t->pipelined) st_wait(sE, t->ev_det[b]);             // Frame.cc:478: while(!isDetectImageFinished())
            TRK_CHECK(sgx_dynamic_mask_batch_dev(S, cap, t->rkeys, t->rn, t->prev_xy, t->F, boxes, nboxes, MB, t->keep, sE));
            TRK_CHECK(sgx_frame_compact_keys_batch_dev(S, cap, t->rkeys, t->rdesc, t->rn, t->keep, have, cf.nfeatures, t->keys[c], t->desc[c], t->n[c], sE));
            // Frame.cc:482-499: this frame's detector results become the "previous frame" state of the next call
            TRK_HIP(hipMemcpyAsync(t->pre_boxes, boxes, (size_t)S * MB * 16, hipMemcpyDeviceToDevice, sE));
            TRK_HIP(hipMemcpyAsync(t->pre_nboxes, nboxes, (size_t)S * 4, hipMemcpyDeviceToDevice, sE));
            TRK_HIP(hipMemcpyAsync(t->pre_have, have, (size_t)S * 4, hipMemcpyDeviceToDevice, sE));
        }
    } else
        TRK_CHECK(sgx_orb_extract_batch_dev(t->ex, d_gray, gray_pitch, S, t->keys[c], t->desc[c], t->n[c], cap, sE));
    TRK_CHECK(sgx_frame_stereo_from_rgbd_batch_dev(S, cap, t->keys[c], t->n[c], d_depth, cf.width, cf.height, cf.depth_map_factor, cf.cam.bf, t->uright[c], t->zdepth[c], sE));
    if (t->pipelined) {
        ev_record(t->ev_extract[c], sE); st_wait(sT, t->ev_extract[c]);
        // "inputs consumed" by the extraction stream's readers; the detector's forward read d_bgr on its own stream and has its own event (ev_det_read[c]): whoever wants to
        // overwrite the inputs waits for both, the extraction stream for neither
        ev_record(t->ev_consumed[c], sE); t->consumed_valid[c] = true;
    }
    // ---- T: Tracking::TrackWithMotionModel (Tracking.cc:906-967)
    if (i > 0) {
        TRK_CHECK(sgx_frame_motion_model_batch_dev(S, Tl, Tll, t->vel_valid, Tc, sT));            // frame 1 has no velocity yet: it starts from the last pose
        TRK_CHECK(sgx_match_project_frame_batch_dev(S, cap, t->keys[c], t->desc[c], t->uright[c], t->n[c], Tc, t->keys[l], t->n[l], t->has[l], t->zero_u8, t->xw[l], t->zero_i4,
                                                    t->desc[l], Tl, &cf.cam, t->scale, t->nlevels, cf.th_projection, 0, 1, t->match, t->nmatch, sT));
        TRK_CHECK(sgx_pose_optimization_batch_dev(S, cap, t->keys[c], t->uright[c], t->n[c], t->match, nullptr, t->xw[l], cap, t->inv_sigma2, t->nlevels, &cf.cam, Tc, t->outlier, t->ninl, sT));
        if (i == 1) TRK_HIP(hipMemsetAsync(t->vel_valid, 1, (size_t)S, sT));
    }
    if (i > 0 && cf.local_map) {
        // ---- Tracking::TrackLocalMap (:969-1013): SearchLocalPoints (isInFrustum + SearchByProjection th = 3) and the second PoseOptimization
        TRK_CHECK(sgx_frame_merge_matches_batch_dev(S, cap, t->n[c], t->match, t->outlier, nullptr, nullptr, nullptr, nullptr, t->cur_mp_obs, nullptr, sT));
        TRK_CHECK(sgx_match_project_local_batch_dev(S, cap, t->keys[c], t->desc[c], t->uright[c], t->n[c], Tc, t->cur_mp_obs, 2 * cap, t->lm_n, t->lm_xw, t->lm_normal, t->lm_min, t->lm_max,
                                                    t->lm_desc, t->lm_obs, t->lm_skip, &cf.cam, t->scale, t->nlevels, t->log_scale, 3.0f, 0.8f, 0.5f, t->match_local, t->nmatch_local, t->in_view, sT));
        TRK_CHECK(sgx_frame_merge_matches_batch_dev(S, cap, t->n[c], t->match, t->outlier, t->match_local, t->xw[l], t->lm_xw, t->merged, nullptr, t->xw_all, sT));
        TRK_CHECK(sgx_pose_optimization_batch_dev(S, cap, t->keys[c], t->uright[c], t->n[c], t->merged, nullptr, t->xw_all, 3 * cap, t->inv_sigma2, t->nlevels, &cf.cam, Tc, t->outlier2, t->ninl2, sT));
    }
    TRK_CHECK(sgx_frame_unproject_batch_dev(S, cap, t->keys[c], t->n[c], t->zdepth[c], Tc, &cf.cam, t->xw[c], t->has[c], sT));
    if (i > 0 && cf.local_map)      // the last frame's points join the local map for the NEXT frames (ring slice (i - 1) % 2 <- frame i - 1)
        TRK_CHECK(sgx_frame_make_map_points_batch_dev(S, cap, (i - 1) % 2, t->keys[l], t->n[l], t->xw[l], t->has[l], t->desc[l], Tl, t->scale, t->nlevels, t->lm_xw, t->lm_normal, t->lm_min,
                                                      t->lm_max, t->lm_desc, t->lm_skip, sT));
    if (t->pipelined) ev_record(t->ev_track[c], sT);
    else { t->last_stream = (sgx_st)caller_stream; ev_record(t->ev_step, t->last_stream); }
    t->Tcw[0] = Tll; t->Tcw[1] = Tc; t->Tcw[2] = Tl;                     // rotate poses: cur -> last, last -> last-last
    t->cur = c; t->frame_idx = i + 1;
    return SGX_OK;
}

// ---- host-input path: the caller fills the pinned staging buffers of a slot (as cv::imread + the TUM loader would, rgbd_tum.cc:114-115), the tracker uploads
// them on its own stream, converts the colour image to gray on the device (Tracking.cc:214-227) and steps.  Two slots.  The upload of a slot is asynchronous: before
// REFILLING a slot the caller asks for it again with sgx_tracker_host_buffers(slot), which returns once the slot's pending H2D copies have completed (or calls
// sgx_tracker_sync).  Refilling on the strength of "two further steps were issued" was unsafe: issuing is not executing.
extern "C" int sgx_tracker_host_buffers(sgx_tracker *t, int slot, uint8_t **bgr, int *bgr_pitch, uint16_t **depth)
{
    if (!t || slot < 0 || slot > 1) return SGX_ERR_INVALID;
    const int S = t->S, W = t->cfg.width, H = t->cfg.height;
    if (!t->h_bgr[0]) {
        t->bgr_pitch = (3 * W + 3) & ~3;
        for (int i = 0; i < 2; i++) {
            t->h_bgr[i] = (uint8_t *)host_alloc((size_t)S * H * t->bgr_pitch); t->h_depth[i] = (uint16_t *)host_alloc((size_t)S * H * W * 2);
            if (!t->h_bgr[i] || !t->h_depth[i]) return SGX_ERR_NOMEM;
            t->pinned.push_back(t->h_bgr[i]); t->pinned.push_back(t->h_depth[i]);
            if (t->alloc(&t->d_bgr[i], (size_t)S * H * t->bgr_pitch) || t->alloc(&t->d_gray[i], (size_t)S * H * W) || t->alloc(&t->d_depth[i], (size_t)S * H * W)) return SGX_ERR_NOMEM;
        }
        if (t->pipelined && st_create(&t->sU, 0)) return SGX_ERR_DEVICE;
        (void)hipDeviceSynchronize();       // zero-fills of the new staging buffers (null stream) before the upload stream touches them
    }
    if (t->up_pending[slot]) { ev_sync(t->ev_up[slot]); t->up_pending[slot] = false; }      // the slot's last upload has left the pinned buffers
    if (bgr) *bgr = t->h_bgr[slot]; if (bgr_pitch) *bgr_pitch = t->bgr_pitch; if (depth) *depth = t->h_depth[slot];
    return SGX_OK;
}

extern "C" int sgx_tracker_step_host(sgx_tracker *t, int slot, int rgb_order)
{
    if (!t || slot < 0 || slot > 1 || !t->h_bgr[0]) return SGX_ERR_INVALID;
    const int S = t->S, W = t->cfg.width, H = t->cfg.height;
    sgx_st sU = t->pipelined ? t->sU : (sgx_st) nullptr;
    // the device staging slot was read by the step issued two calls ago, on the extraction AND the detector stream: ev_consumed + ev_det_read (ev_extract alone does not
    // cover the detector when that step ran without the mask's detector wait — first frame, dynamic_mask == 0)
    if (t->pipelined && t->frame_idx >= 2) { const int p = (t->frame_idx - 2) % 3; st_wait(sU, t->ev_consumed[p]); if (t->det_read_valid[p]) st_wait(sU, t->ev_det_read[p]); }
    TRK_HIP(hipMemcpyAsync(t->d_bgr[slot], t->h_bgr[slot], (size_t)S * H * t->bgr_pitch, hipMemcpyHostToDevice, sU));
    TRK_HIP(hipMemcpyAsync(t->d_depth[slot], t->h_depth[slot], (size_t)S * H * W * 2, hipMemcpyHostToDevice, sU));
    ev_record(t->ev_up[slot], sU); t->up_pending[slot] = true;          // non-pipelined too: the copies out of pinned memory on the null stream are asynchronous to the host
    TRK_CHECK(sgx_frame_gray_from_color_batch_dev(S, W, H, t->d_bgr[slot], t->bgr_pitch, 3, rgb_order ? 0 : 1, t->d_gray[slot], W, sU));
    return sgx_tracker_step_dev(t, t->d_gray[slot], W, t->d_depth[slot], t->det ? t->d_bgr[slot] : nullptr, t->bgr_pitch, sU);
}

// Blocks the host until the device has read the INPUT images of the step issued `steps_back` calls ago (0 = the last one; at most 2): the caller may then rewrite
// those buffers from the host.  (Stream-ordered writers need nothing: see sgx_tracker_step_dev.)
extern "C" int sgx_tracker_wait_inputs(sgx_tracker *t, int steps_back)
{
    if (!t || steps_back < 0 || steps_back > 2) return SGX_ERR_INVALID;
    if (t->frame_idx - 1 - steps_back < 0) return SGX_OK;
    if (!t->pipelined) { TRK_HIP(hipDeviceSynchronize()); return SGX_OK; }
    const int c = (t->frame_idx - 1 - steps_back) % 3;
    if (t->consumed_valid[c]) ev_sync(t->ev_consumed[c]);
    if (t->det_read_valid[c]) ev_sync(t->ev_det_read[c]);
    return SGX_OK;
}

extern "C" int sgx_tracker_sync(sgx_tracker *t)
{
    if (!t) return SGX_ERR_INVALID;
    if (t->pipelined) { st_sync(t->sE); st_sync(t->sT); st_sync(t->sD); if (t->sU) st_sync(t->sU); }
    else TRK_HIP(hipDeviceSynchronize());
    t->up_pending[0] = t->up_pending[1] = false;
    return SGX_OK;
}

// results of the frame tracked last (synchronises): Tcw (S x 16), keypoints after the mask, motion-model matches / inliers, local-map matches / inliers of the
// second PoseOptimization, keypoints before the mask, findFundamentalMat success and RANSAC iteration count.  Any pointer may be NULL.
extern "C" int sgx_tracker_read(sgx_tracker *t, float *Tcw, int32_t *nkeys, int32_t *nmatches, int32_t *ninliers, int32_t *nmatches_local, int32_t *ninliers2, int32_t *nkeys_raw,
                                int32_t *f_ok, int32_t *f_stats)
{
    if (!t) return SGX_ERR_INVALID;
    TRK_CHECK(sgx_tracker_sync(t));
    const size_t S = (size_t)t->S;
    if (Tcw) TRK_HIP(hipMemcpy(Tcw, t->Tcw[1], S * 64, hipMemcpyDeviceToHost));
    if (nkeys) TRK_HIP(hipMemcpy(nkeys, t->n[t->cur], S * 4, hipMemcpyDeviceToHost));
    if (nmatches) TRK_HIP(hipMemcpy(nmatches, t->nmatch, S * 4, hipMemcpyDeviceToHost));
    if (ninliers) TRK_HIP(hipMemcpy(ninliers, t->ninl, S * 4, hipMemcpyDeviceToHost));
    if (nmatches_local) TRK_HIP(hipMemcpy(nmatches_local, t->nmatch_local, S * 4, hipMemcpyDeviceToHost));
    if (ninliers2) TRK_HIP(hipMemcpy(ninliers2, t->ninl2, S * 4, hipMemcpyDeviceToHost));
    if (nkeys_raw) TRK_HIP(hipMemcpy(nkeys_raw, t->rn, S * 4, hipMemcpyDeviceToHost));
    if (f_ok) TRK_HIP(hipMemcpy(f_ok, t->f_ok, S * 4, hipMemcpyDeviceToHost));
    if (f_stats) TRK_HIP(hipMemcpy(f_stats, t->f_stats, S * 16, hipMemcpyDeviceToHost));
    return SGX_OK;
}

// device copy of the pose of the frame just tracked (S x 16 floats), ordered on the tracking stream: trajectory files / ATE without a host sync per frame
extern "C" int sgx_tracker_snapshot_pose_dev(sgx_tracker *t, float *d_out)
{
    if (!t || !d_out) return SGX_ERR_INVALID;
    TRK_HIP(hipMemcpyAsync(d_out, t->Tcw[1], (size_t)t->S * 64, hipMemcpyDeviceToDevice, t->pipelined ? t->sT : t->last_stream));      // non-pipelined: the step ran on the caller's stream
    return SGX_OK;
}

// device copy of one stream's person boxes of the frame just issued (max_boxes x 4 floats + count), ordered on the detector stream (oracle-chain comparison)
extern "C" int sgx_tracker_snapshot_boxes_dev(sgx_tracker *t, int stream_index, float *d_boxes, int32_t *d_nboxes)
{
    if (!t || !d_boxes || !d_nboxes || stream_index < 0 || stream_index >= t->S || t->frame_idx < 1) return SGX_ERR_INVALID;
    const int b = (t->frame_idx - 1) & 1;
    sgx_st sD = t->pipelined ? t->sD : t->last_stream;
    TRK_HIP(hipMemcpyAsync(d_boxes, t->det_boxes[b] + (size_t)stream_index * t->MB * 4, (size_t)t->MB * 16, hipMemcpyDeviceToDevice, sD));
    TRK_HIP(hipMemcpyAsync(d_nboxes, t->det_nb[b] + stream_index, 4, hipMemcpyDeviceToDevice, sD));
    return SGX_OK;
}

// BASELINE config 5: the per-frame records {n, cv::KeyPoint[cap], descriptors[cap][32], Tcw} of the frame just tracked, packed by ONE kernel into
// d_records (S x sgx_tracker_record_bytes) on `stream`, which first waits for the frame's tracking event; the frame slot is not reused before the pack ran.
extern "C" int sgx_tracker_pack_records_dev(sgx_tracker *t, uint8_t *d_records, void *stream)
{
    if (!t || !d_records || t->frame_idx < 1 || ((uintptr_t)d_records & 3)) return SGX_ERR_INVALID;
    const int c = t->cur, words = sgx_tracker_record_bytes(t) / 4;
    sgx_st st = (sgx_st)stream;
    if (t->pipelined) st_wait(st, t->ev_track[c]);
    else if (st != t->last_stream) st_wait(st, t->ev_step);               // non-pipelined: the step ran on the caller's stream, `stream` may be another (non-blocking) one
    SGX_LAUNCH(k_pack_frame_records, dim3(t->S), dim3(256), st, t->cap, t->n[c], (const uint32_t *)t->keys[c], (const uint32_t *)t->desc[c], (const uint32_t *)t->Tcw[1], (uint32_t *)d_records, words);
    TRK_HIP(hipGetLastError());
    if (t->pipelined) { ev_record(t->ev_pack[c], st); t->pack_pending[c] = true; }
    return SGX_OK;
}

// device pointers of the frame tracked last (valid until two more steps have been issued): tests and callers that read keypoints / descriptors in place
extern "C" int sgx_tracker_frame_dev(sgx_tracker *t, const int32_t **d_n, const sgx_keypoint **d_keys, const uint8_t **d_desc, const float **d_Tcw, const float **d_xw, const uint8_t **d_has)
{
    if (!t || t->frame_idx < 1) return SGX_ERR_INVALID;
    const int c = t->cur;
    if (d_n) *d_n = t->n[c]; if (d_keys) *d_keys = t->keys[c]; if (d_desc) *d_desc = t->desc[c]; if (d_Tcw) *d_Tcw = t->Tcw[1]; if (d_xw) *d_xw = t->xw[c]; if (d_has) *d_has = t->has[c];
    return SGX_OK;
}

extern "C" sgx_orb *sgx_tracker_extractor(sgx_tracker *t) { return t ? t->ex : nullptr; }
