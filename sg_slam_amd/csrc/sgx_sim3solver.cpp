// sgx_sim3solver.cpp — host side of Sim3Solver (src/sg-slam/src/Sim3Solver.cc) behind the C ABI: the solver object (correspondences, RANSAC state, best model),
// SetRansacParameters and the resumable iterate(); the hypotheses of a call run on the device (sgx_sim3solver_kernels.h).
#include "sgx_sim3solver_kernels.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

struct sgx_sim3_solver {
    int N = 0, fix_scale = 0;
    double prob = 0.99; int minInliers = 6, maxIts = 300, nIterations = 0, nBestInliers = 0;
    float K1[4], K2[4];
    float *dX1 = nullptr, *dX2 = nullptr, *dP1 = nullptr, *dP2 = nullptr, *dE1 = nullptr, *dE2 = nullptr, *dHyp = nullptr; int *dDraws = nullptr, *dRes = nullptr; uint8_t *dInl = nullptr;
    float bestT12[16], bestR[9], bestt[3], bests = 1.f;
    // glibc rand() replica (TYPE_3: r[i] = r[i - 3] + r[i - 31]) for callers that do not pass their own rand() values
    int32_t gr[34]; int gf = 3, gb = 0;
    void gsrand(unsigned seed)
    {
        int32_t word = seed ? (int32_t)seed : 1; gr[0] = word;
        for (int i = 1; i < 31; i++) { const long hi = word / 127773, lo = word % 127773; long w = 16807 * lo - 2836 * hi; if (w < 0) w += 2147483647; word = (int32_t)w; gr[i] = word; }
        gf = 3; gb = 0;
        for (int i = 0; i < 310; i++) { gr[gf] = (int32_t)((uint32_t)gr[gf] + (uint32_t)gr[gb]); gf = (gf + 1) % 31; gb = (gb + 1) % 31; }
    }
    int32_t grand() { gr[gf] = (int32_t)((uint32_t)gr[gf] + (uint32_t)gr[gb]); const int32_t o = (int32_t)(((uint32_t)gr[gf]) >> 1); gf = (gf + 1) % 31; gb = (gb + 1) % 31; return o; }
    ~sgx_sim3_solver() { for (void *p : { (void *)dX1, (void *)dX2, (void *)dP1, (void *)dP2, (void *)dE1, (void *)dE2, (void *)dHyp, (void *)dDraws, (void *)dRes, (void *)dInl }) if (p) (void)hipFree(p); }
};

extern "C" int sgx_sim3_solver_set_ransac_parameters(sgx_sim3_solver *s, double probability, int min_inliers, int max_iterations)
{
    if (!s) return SGX_ERR_INVALID;
    if (!(probability > 0 && probability < 1) || min_inliers < 1 || max_iterations < 1) return SGX_ERR_INVALID;
    s->prob = probability; s->minInliers = min_inliers; s->maxIts = max_iterations;                 // Sim3Solver.cc:113-138
    s->nIterations = 0;
    if (s->N <= 0) return SGX_OK;                                 // no correspondences: iterate() reports bNoMore at once (N < minInliers); the reference divides by N here
    const float epsilon = (float)s->minInliers / s->N;
    int nIterations;
    if (s->minInliers == s->N) nIterations = 1;
    else nIterations = (int)ceil(log(1 - s->prob) / log(1 - pow(epsilon, 3)));
    const int m = nIterations < s->maxIts ? nIterations : s->maxIts;
    s->maxIts = m > 1 ? m : 1;
    s->nIterations = 0;
    return SGX_OK;
}

extern "C" int sgx_sim3_solver_create(int n, const float *x3dc1, const float *x3dc2, const float *max_err1, const float *max_err2, const float *K1, const float *K2, int fix_scale,
                                      unsigned rand_seed, sgx_sim3_solver **out)
{
    if (!out || n < 0 || !K1 || !K2 || (n > 0 && (!x3dc1 || !x3dc2 || !max_err1 || !max_err2))) return SGX_ERR_INVALID;
    sgx_sim3_solver *s = new sgx_sim3_solver;
    s->N = n; s->fix_scale = fix_scale ? 1 : 0; memcpy(s->K1, K1, 16); memcpy(s->K2, K2, 16);
    memset(s->bestT12, 0, sizeof s->bestT12); memset(s->bestR, 0, sizeof s->bestR); memset(s->bestt, 0, sizeof s->bestt);
    s->gsrand(rand_seed);
    const size_t m = (size_t)(n > 0 ? n : 1);
    std::vector<float> p1(2 * m), p2(2 * m);
    for (int i = 0; i < n; i++) {                                // FromCameraToImage :407-425
        const float iz1 = 1 / x3dc1[3 * i + 2], iz2 = 1 / x3dc2[3 * i + 2];
        p1[2 * (size_t)i] = K1[0] * (x3dc1[3 * i] * iz1) + K1[2]; p1[2 * (size_t)i + 1] = K1[1] * (x3dc1[3 * i + 1] * iz1) + K1[3];
        p2[2 * (size_t)i] = K2[0] * (x3dc2[3 * i] * iz2) + K2[2]; p2[2 * (size_t)i + 1] = K2[1] * (x3dc2[3 * i + 1] * iz2) + K2[3];
    }
    bool ok = hipMalloc((void **)&s->dX1, 12 * m) == hipSuccess && hipMalloc((void **)&s->dX2, 12 * m) == hipSuccess && hipMalloc((void **)&s->dP1, 8 * m) == hipSuccess &&
              hipMalloc((void **)&s->dP2, 8 * m) == hipSuccess && hipMalloc((void **)&s->dE1, 4 * m) == hipSuccess && hipMalloc((void **)&s->dE2, 4 * m) == hipSuccess &&
              hipMalloc((void **)&s->dHyp, sizeof(float) * SGX_S3_HYP * SGX_S3_MAXIT) == hipSuccess && hipMalloc((void **)&s->dDraws, 12 * SGX_S3_MAXIT) == hipSuccess &&
              hipMalloc((void **)&s->dRes, 16) == hipSuccess && hipMalloc((void **)&s->dInl, m) == hipSuccess;
    if (ok && n > 0)
        ok = hipMemcpy(s->dX1, x3dc1, 12 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(s->dX2, x3dc2, 12 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(s->dP1, p1.data(), 8 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(s->dP2, p2.data(), 8 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(s->dE1, max_err1, 4 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(s->dE2, max_err2, 4 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { delete s; return SGX_ERR_NOMEM; }
    if (n > 0) sgx_sim3_solver_set_ransac_parameters(s, 0.99, 6, 300);             // the constructor ends with SetRansacParameters() (:110)
    *out = s;
    return SGX_OK;
}

extern "C" void sgx_sim3_solver_destroy(sgx_sim3_solver *s) { delete s; }

extern "C" int sgx_sim3_solver_iterate(sgx_sim3_solver *s, int n_iterations, const int32_t *rand_draws, float *T12, int32_t *no_more, uint8_t *inliers, int32_t *n_inliers,
                                       int32_t *found, int32_t *iterations_run)
{
    if (!s || !T12 || !no_more || !n_inliers || !found || (s->N > 0 && !inliers) || n_iterations < 0) return SGX_ERR_INVALID;
    *no_more = 0; *n_inliers = 0; *found = 0; if (iterations_run) *iterations_run = 0;
    for (int i = 0; i < s->N; i++) inliers[i] = 0;
    if (s->N < s->minInliers) { *no_more = 1; return SGX_OK; }                       // :146-150
    if (s->N < 3) return SGX_ERR_INVALID;                                            // the reference would index an empty vAvailableIndices
    int run_total = 0;
    while (s->nIterations < s->maxIts && run_total < n_iterations) {                 // chunks of SGX_S3_MAXIT hypotheses
        int n_it = n_iterations - run_total; if (n_it > s->maxIts - s->nIterations) n_it = s->maxIts - s->nIterations; if (n_it > SGX_S3_MAXIT) n_it = SGX_S3_MAXIT;
        std::vector<int32_t> draws((size_t)3 * n_it);
        // the reference draws from the process-global rand() as it goes; a caller that shares that stream passes the values, otherwise the solver's own replica supplies them.
        // Draws of iterations that are not run (early success) are handed back below.
        int32_t gr0[34]; int gf0 = s->gf, gb0 = s->gb; memcpy(gr0, s->gr, sizeof gr0);
        for (int i = 0; i < 3 * n_it; i++) draws[(size_t)i] = rand_draws ? rand_draws[3 * run_total + i] : s->grand();
        SGX_CHECK_HIP(hipMemcpy(s->dDraws, draws.data(), draws.size() * 4, hipMemcpyHostToDevice));
        SgxS3Args A; memset(&A, 0, sizeof A);
        A.N = s->N; A.fix_scale = s->fix_scale; A.n_iter = n_it; A.min_inliers = s->minInliers; A.best_in = s->nBestInliers;
        A.X1 = s->dX1; A.X2 = s->dX2; A.P1im1 = s->dP1; A.P2im2 = s->dP2; A.maxErr1 = s->dE1; A.maxErr2 = s->dE2; memcpy(A.K1, s->K1, 16); memcpy(A.K2, s->K2, 16);
        A.draws = s->dDraws; A.hyp = s->dHyp; A.result = s->dRes; A.inl = s->dInl;
        SGX_LAUNCH(k_sim3_ransac, dim3(1), dim3(256), (sgx_stream_t)0, A);
        SGX_CHECK_HIP(hipGetLastError());
        int res[4];
        SGX_CHECK_HIP(hipMemcpy(res, s->dRes, 16, hipMemcpyDeviceToHost));
        const int fnd = res[0], run = res[1], best_h = res[2];
        if (!rand_draws && run < n_it) {                                             // give back the draws of the iterations that did not run
            memcpy(s->gr, gr0, sizeof gr0); s->gf = gf0; s->gb = gb0;
            for (int i = 0; i < 3 * run; i++) (void)s->grand();
        }
        s->nIterations += run; run_total += run;
        if (best_h >= 0) {
            float h[SGX_S3_HYP];
            SGX_CHECK_HIP(hipMemcpy(h, s->dHyp + (size_t)best_h * SGX_S3_HYP, sizeof h, hipMemcpyDeviceToHost));
            memcpy(s->bestT12, h, 64); memcpy(s->bestR, h + 32, 36); memcpy(s->bestt, h + 41, 12); s->bests = h[44]; s->nBestInliers = res[3];
        }
        if (fnd >= 0) {
            SGX_CHECK_HIP(hipMemcpy(inliers, s->dInl, (size_t)s->N, hipMemcpyDeviceToHost));
            memcpy(T12, s->bestT12, 64); *n_inliers = res[3]; *found = 1;
            if (iterations_run) *iterations_run = run_total;
            return SGX_OK;
        }
    }
    if (iterations_run) *iterations_run = run_total;
    if (s->nIterations >= s->maxIts) *no_more = 1;
    return SGX_OK;
}

extern "C" int sgx_sim3_solver_get_estimate(const sgx_sim3_solver *s, float *R12, float *t12, float *scale, int32_t *max_iterations)
{
    if (!s) return SGX_ERR_INVALID;
    if (R12) memcpy(R12, s->bestR, 36); if (t12) memcpy(t12, s->bestt, 12); if (scale) *scale = s->bests; if (max_iterations) *max_iterations = s->maxIts;
    return SGX_OK;
}
