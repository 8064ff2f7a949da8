// sgx_prof.cpp — per-kernel-class HIP-event timing (events recorded on the caller's stream around each launch).
#include "sgx_prof.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <vector>
#include <mutex>

static int g_on = 0;
static const char *g_names[SGX_K_COUNT] = { "pyramid_resize", "fast_cells", "octree", "orient_desc", "stereo_from_rgbd",
                                            "motion_model", "match_project_frame", "pose_opt", "unproject", "match_project_local", "map_point_glue", "dynamic_mask", "lk_pyramid", "lk_track", "fm_ransac", "det_forward", "det_output",
                                            "ba_linearize", "ba_schur", "ba_solve", "ba_update" };
#ifndef SGX_EMU
static std::vector<hipEvent_t> g_a[SGX_K_COUNT], g_b[SGX_K_COUNT];
static int g_used[SGX_K_COUNT];
// the entry points are called from the Tracking, Detector2D and LocalMapping threads: the event lists are process-wide, so every access takes this
// lock.  A begin/end pair of one class must not interleave with another thread's pair of the SAME class (each class belongs to one pipeline stage, hence one thread).
static std::mutex g_mu;
void sgx_prof_begin(int k, sgx_stream_t st)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used[k] == (int)g_a[k].size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); g_a[k].push_back(a); g_b[k].push_back(b); }
    (void)hipEventRecord(g_a[k][g_used[k]], st);
}
void sgx_prof_end(int k, sgx_stream_t st)
{
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used[k] >= (int)g_a[k].size()) return;                      // end without a begin (profiling switched on in between)
    (void)hipEventRecord(g_b[k][g_used[k]], st);
    g_used[k]++;
}
#else
void sgx_prof_begin(int, sgx_stream_t) {}
void sgx_prof_end(int, sgx_stream_t) {}
#endif

extern "C" int sgx_profile_enable(int on) { g_on = on ? 1 : 0; return SGX_OK; }
extern "C" int sgx_profile_num_classes(void) { return SGX_K_COUNT; }
extern "C" const char *sgx_profile_class_name(int k) { return (k >= 0 && k < SGX_K_COUNT) ? g_names[k] : ""; }

extern "C" int sgx_profile_read(float *ms, int32_t *launches, int reset)
{
    if (!ms || !launches) return SGX_ERR_INVALID;
    for (int k = 0; k < SGX_K_COUNT; k++) { ms[k] = 0.f; launches[k] = 0; }
#ifndef SGX_EMU
    if (hipDeviceSynchronize() != hipSuccess) return SGX_ERR_DEVICE;
    std::lock_guard<std::mutex> lk(g_mu);
    for (int k = 0; k < SGX_K_COUNT; k++) {
        for (int i = 0; i < g_used[k]; i++) { float t = 0.f; if (hipEventElapsedTime(&t, g_a[k][i], g_b[k][i]) != hipSuccess) return SGX_ERR_DEVICE; ms[k] += t; }
        launches[k] = g_used[k];
        if (reset) g_used[k] = 0;
    }
#endif
    return SGX_OK;
}
