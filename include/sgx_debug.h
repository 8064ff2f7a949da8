/* sgx_debug.h — TEST / TUNING TAPS of libsgx.  NOT part of the product: the default build of sg_slam_amd/libsgx.so exports none of these symbols and reads no
 * SGX_* environment variable (every sgx_getenv() in csrc/ is a constant NULL there).  They exist only in builds with -DSGX_DEBUG_TAPS:
 *   tests/taps/libsgx_taps.so   (hipcc, gfx950; `make -C sg_slam_amd/csrc taps`)  — the GPU tests that read intermediate blobs / pyramid levels / force a plan, and the A/B tools
 *   tests/emu/libsgx_emu.so     (g++, -DSGX_EMU)                                   — the kernel-logic emulator of the CPU tests
 * Same sources, same kernels; the taps only add read-backs and plan selection.  Every sgx_*_debug_set_* setting is PER CALLING THREAD (thread_local): it affects the
 * next create / call made by the same thread only. */
#ifndef SGX_DEBUG_H
#define SGX_DEBUG_H
#include "sgx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* test/diagnostic taps (synchronous, host destination) */
int sgx_orb_debug_level_geometry(const sgx_orb *h, int level, int32_t *w, int32_t *hgt, int32_t *stride);
int sgx_orb_debug_set_unfused_pyramid(int on);   /* test tap: 1 = per-level k_resize launches instead of the fused pyramid kernel (must give identical bytes) */
int sgx_orb_debug_read_level(sgx_orb *h, int frame, int level, uint8_t *dst /* w*h, tight */);
/* candidates of (frame, level) after per-cell FAST+NMS, unordered: x,y relative to the
 * (16,16) border origin exactly as pushed at ORBextractor.cc:823-825; returns count in *n */
int sgx_orb_debug_read_candidates(sgx_orb *h, int frame, int level, int32_t *x, int32_t *y, int32_t *score, int cap, int *n);
/* Harness helper (bench.py / tests), NOT a reference entry point: the previous-frame position of every keypoint under a per-frame affine flow
 * (prev = A * (x, y, 1), A = 6 floats), optionally displaced by shift[2] inside the frame's first box.  Stands in for cv::calcOpticalFlowPyrLK
 * (Frame.cc:445, tier N1) when the dynamic-feature mask is exercised on synthetic streams whose flow is known exactly. */
int sgx_debug_flow_affine_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_A, const float *d_shift, const float *d_boxes,
                                    int max_boxes, float *d_prev_xy, void *stream);
int sgx_pose_opt_debug_set_threads(int threads_per_frame);   /* tuning / test tap: 0 = default (256 threads = four waves per frame), 64 = one wave per frame, 256 */
int sgx_ba_debug_last_plan(int32_t plan[4]);    /* test tap: solver plan of the last bundle adjustment of this process: { 0 dense / 1 envelope, column steps of branch A, of branch B (0 = one branch), unknowns of their separator } */
int sgx_ba_debug_set_init(int mode);               /* test tap: initialisation of the reduced camera system when the envelope solver runs: 0 only the solver's tiles (default), 1 the whole matrix, 2 the whole matrix NaN first, then the tiles */
int sgx_ba_debug_set_jobs(int host);               /* test tap: 1 = the bundle adjustments of this thread build their Schur job list on the host (A/B arm of the device builder), 0 = on the device (default) */
int sgx_ba_debug_set_solver(int mode);             /* test / tuning tap: reduced-camera-system solver of the bundle adjustments: -1 default (SGX_BA_SOLVER or auto), 0 auto, 1 dense blocked Cholesky, 2 envelope solver */
int sgx_det_debug_read_blob(sgx_det *h, const char *blob_name, int image, float *dst, int cap, int *n);
int sgx_flow_debug_read_level(sgx_flow *h, int slot, int frame, int level, uint8_t *img);     /* test tap: pyramid level, w*h tight */
int sgx_flow_debug_level_size(const sgx_flow *h, int level, int32_t *w, int32_t *hgt);
int sgx_debug_corun_bf16(int blocks, int iters, int launches, void *stream);      /* test tap: `launches` launches of a kernel that only issues bf16 matrix products, asynchronous on `stream` (co-runner of the packed-fp32 regression test) */
/* test tap: DetectionOutput + detect() filtering alone on caller-supplied head outputs (host arrays: loc batch x num_priors x 4, conf batch x num_priors x num_class) */
int sgx_det_debug_detection_output(sgx_det *h, const float *loc, const float *conf, int batch, sgx_det_result *results);
/* test / tuning taps.  Every sgx_*_debug_set_* setting is PER CALLING THREAD (thread_local): it affects the next create / call made by the same thread only, so the taps
 * cannot leak into the Tracking, Detector2D or LocalMapping thread of a host program; the SGX_* environment switches are read once and never written.
 * set_fusion(0) makes the NEXT sgx_det_create build the unfused plan (one kernel per ncnn layer, every blob
 * materialised) — the fused plan (default) must reproduce it bit for bit.  time_ops: HIP-event time per plan step (ms[0] = pre-processing). */
int sgx_det_debug_set_fusion(int on);
int sgx_det_debug_set_irb(int on);                 /* the NEXT sgx_det_create: matrix-core inverted-residual block kernels (sgx_det_irb.h) 0 off, 1 on the shapes where they beat the per-layer kernels, 2 on every supported shape, -1 = default (1, or SGX_DET_IRB); bit-identical either way */
int sgx_det_debug_set_block_fusion(int on);        /* 1: the NEXT sgx_det_create also fuses every expand -> depthwise -> project triple into one kernel (bit-identical; opt-in: slower at batch 256) */
int sgx_det_debug_set_gemm(int mode);
int sgx_det_debug_set_legacy_kernels(int on);      /* 1: run the simple reference kernels (one thread per output / 64x64 GEMM tile) instead of the tuned ones */
int sgx_det_debug_time_ops(sgx_det *h, const uint8_t *d_img, int pitch, int batch, int reps, float *ms, int cap, int *nops);
int sgx_det_debug_run_step(sgx_det *h, const uint8_t *d_img, int pitch, int batch, int step, int reps, void *stream);      /* plan step `step` launched reps times on `stream`, asynchronously (interference experiments) */
/* run the octree-distribution kernel alone on packed candidates (x | y<<12 | score<<24, coordinates
 * relative to the (16,16) border origin) for `level`; returns the selected packed entries in list order */
int sgx_orb_debug_run_octree(sgx_orb *h, int level, const uint32_t *packed, int n, uint32_t *out_sel, int cap, int *nsel);

#ifdef __cplusplus
}
#endif
#endif
