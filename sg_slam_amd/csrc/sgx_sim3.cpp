// sgx_sim3.cpp — host side of Optimizer::OptimizeSim3 (include/sgx.h: sgx_optimize_sim3).  Reference: src/sg-slam/src/Optimizer.cc:1046-1257.
#include "sgx_sim3_kernels.h"
#include "sgx_stage.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <stdio.h>
#include <string.h>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

extern "C" int sgx_optimize_sim3(int n, const float *p1c, const float *p2c, const float *obs1, const float *obs2, const float *info1, const float *info2,
                                 const float *K1, const float *K2, double *S12, float th2, int fix_scale, uint8_t *inlier, int32_t *iterations, int32_t *n_inliers)
{
    if (n < 0 || !K1 || !K2 || !S12 || !n_inliers || (n > 0 && (!p1c || !p2c || !obs1 || !obs2 || !info1 || !info2 || !inlier))) return SGX_ERR_INVALID;
    *n_inliers = 0;
    if (iterations) { iterations[0] = 0; iterations[1] = 0; }
    if (n < 10) {                                          // with fewer than 10 correspondences the reference optimises 5 iterations and then returns 0 without reading the estimate back:
        for (int i = 0; i < n; i++) inlier[i] = 1;          // nothing observable changes except vpMatches1 entries NULLed by the first check — run the kernel only when n >= 1 to get those flags
        if (n == 0) return SGX_OK;
    }
    SgxSim3Args A; memset(&A, 0, sizeof A);
    A.n = n; A.fix_scale = fix_scale ? 1 : 0; A.th2 = th2;
    for (int i = 0; i < 4; i++) { A.K1[i] = K1[i]; A.K2[i] = K2[i]; }
    SgxStaged b[12]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, p1c, (size_t)n * 12); PUT(1, p2c, (size_t)n * 12); PUT(2, obs1, (size_t)n * 8); PUT(3, obs2, (size_t)n * 8); PUT(4, info1, (size_t)n * 4); PUT(5, info2, (size_t)n * 4);
    PUT(6, S12, 64); PUT(7, nullptr, (size_t)n * 32); PUT(8, nullptr, (size_t)n); PUT(9, nullptr, 8); PUT(10, nullptr, 4);
#undef PUT
    A.p1c = (const float *)b[0].p; A.p2c = (const float *)b[1].p; A.obs1 = (const float *)b[2].p; A.obs2 = (const float *)b[3].p; A.info1 = (const float *)b[4].p; A.info2 = (const float *)b[5].p;
    A.S12 = (double *)b[6].p; A.err = (double *)b[7].p; A.inlier = (uint8_t *)b[8].p; A.iters = (int *)b[9].p; A.nin = (int *)b[10].p;
    SGX_LAUNCH(k_optimize_sim3, dim3(1), dim3(256), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    int nin = 0, its[2] = { 0, 0 };
    SGX_CHECK_HIP(hipMemcpy(&nin, A.nin, 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(its, A.iters, 8, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(inlier, A.inlier, (size_t)n, hipMemcpyDeviceToHost));
    if (nin > 0 || its[1] > 0) SGX_CHECK_HIP(hipMemcpy(S12, A.S12, 64, hipMemcpyDeviceToHost));     // the "fewer than 10 survivors" exit leaves g2oS12 untouched
    *n_inliers = nin;
    if (iterations) { iterations[0] = its[0]; iterations[1] = its[1]; }
    return SGX_OK;
}
