// sgx_sim3solver_kernels.h — Sim3Solver (src/sg-slam/src/Sim3Solver.cc), the RANSAC initialiser of LoopClosing::ComputeSim3 (LoopClosing.cc:274-301):
//   iterate :140-208, ComputeCentroid :216-225, ComputeSim3 (Horn 1987) :228-337, CheckInliers :340-365, Project :383-405.
// One workgroup per call: every iteration of the call is an independent hypothesis — thread h draws its three correspondences (the swap-and-pop of vAvailableIndices
// resolved in closed form), runs Horn's closed form (centroids, M = Pr2 Pr1^T, the 4 x 4 symmetric N, its dominant eigenvector by cv::eigen's Jacobi sweep, Rodrigues,
// scale, translation, T12 / T21) in the cv::Mat arithmetic of the reference; then all threads count the inliers of all hypotheses ((hypothesis, point) pairs dealt flat);
// thread 0 replays the sequential accept rule (`>=` best so far, return at the first hypothesis with more than mRansacMinInliers inliers).
#pragma once
#include "sgx_match_common.h"
#include <float.h>

#define SGX_S3_MAXIT 512
#define SGX_S3_HYP 48                 /* floats per hypothesis: T12 16 | T21 16 | R 9 | t 3 | s 1 | pad */

struct SgxS3Args {
    int N, fix_scale, n_iter, min_inliers, best_in;
    const float *X1, *X2, *P1im1, *P2im2, *maxErr1, *maxErr2;
    float K1[4], K2[4];
    const int *draws;                  // 3 raw rand() values per iteration
    float *hyp;                        // n_iter x SGX_S3_HYP
    int *result;                       // found iteration (-1), iterations run, best iteration of this call (-1), its inlier count
    uint8_t *inl;                      // inlier mask of the best iteration of this call
};

SGX_DEV float sgx_gemm_small3(const float *arow, const float *b, int bstep, double alpha, double beta, float c)
{
    const float t = arow[0] * b[0] + arow[1] * b[bstep] + arow[2] * b[2 * bstep];
    return (float)((double)t * alpha + (double)c * beta);
}

// cv::eigen on a symmetric float matrix = JacobiImpl_<float> (OpenCV lapack.cpp): eigenvalues descending, eigenvectors as rows
SGX_DEV void sgx_jacobi4(float *A, float *W, float *V)
{
    const int n = 4; const float eps = FLT_EPSILON;
    int indR[4], indC[4], i, k, m; float mv = 0;
    for (i = 0; i < n; i++) for (k = 0; k < n; k++) V[i * n + k] = i == k ? 1.f : 0.f;
    for (k = 0; k < n; k++) {
        W[k] = A[(n + 1) * k];
        if (k < n - 1) { for (m = k + 1, mv = fabsf(A[n * k + m]), i = k + 2; i < n; i++) { const float val = fabsf(A[n * k + i]); if (mv < val) mv = val, m = i; } indR[k] = m; }
        if (k > 0) { for (m = 0, mv = fabsf(A[k]), i = 1; i < k; i++) { const float val = fabsf(A[n * i + k]); if (mv < val) mv = val, m = i; } indC[k] = m; }
    }
    for (int iters = 0; iters < n * n * 30; iters++) {
        for (k = 0, mv = fabsf(A[indR[0]]), i = 1; i < n - 1; i++) { const float val = fabsf(A[n * i + indR[i]]); if (mv < val) mv = val, k = i; }
        int l = indR[k];
        for (i = 1; i < n; i++) { const float val = fabsf(A[n * indC[i] + i]); if (mv < val) mv = val, k = indC[i], l = i; }
        const float p = A[n * k + l];
        if (fabsf(p) <= eps) break;
        const float y = (float)((double)(W[l] - W[k]) * 0.5);
        float t = fabsf(y) + hypotf(p, y);
        float s = hypotf(p, t);
        const float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t; W[l] += t;
        float a0, b0;
#define SGX_ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (i = 0; i < k; i++) SGX_ROT(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; i++) SGX_ROT(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; i++) SGX_ROT(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; i++) SGX_ROT(V[n * k + i], V[n * l + i]);
#undef SGX_ROT
        for (int j = 0; j < 2; j++) {
            const int idx = j == 0 ? k : l;
            if (idx < n - 1) { for (m = idx + 1, mv = fabsf(A[n * idx + m]), i = idx + 2; i < n; i++) { const float val = fabsf(A[n * idx + i]); if (mv < val) mv = val, m = i; } indR[idx] = m; }
            if (idx > 0) { for (m = 0, mv = fabsf(A[idx]), i = 1; i < idx; i++) { const float val = fabsf(A[n * i + idx]); if (mv < val) mv = val, m = i; } indC[idx] = m; }
        }
    }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) { const float tw = W[m]; W[m] = W[k]; W[k] = tw; for (i = 0; i < n; i++) { const float tv = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = tv; } }
    }
}

// cv::Rodrigues(vector -> matrix): double arithmetic, float result
SGX_DEV void sgx_rodrigues(const float *v, float *R)
{
    const double rx0 = v[0], ry0 = v[1], rz0 = v[2];
    const double theta = sqrt(rx0 * rx0 + ry0 * ry0 + rz0 * rz0);
    if (theta < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4) == 0 ? 1.f : 0.f; return; }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    const double rx = rx0 * itheta, ry = ry0 * itheta, rz = rz0 * itheta;
    const double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
    const double rx_[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    for (int i = 0; i < 9; i++) R[i] = (float)(c * ((i % 4) == 0 ? 1. : 0.) + c1 * rrt[i] + s * rx_[i]);
}

// ComputeSim3 :228-337; P1, P2: 3 x 3 row-major, column i = point i.  out: SGX_S3_HYP floats
SGX_DEV void sgx_compute_sim3(const float *P1, const float *P2, int fix_scale, float *out)
{
    float Pr1[9], Pr2[9], O1[3], O2[3];
    const float third = (float)(1.0 / 3);
    for (int r = 0; r < 3; r++) {
        O1[r] = (P1[3 * r] + P1[3 * r + 1] + P1[3 * r + 2]) * third;
        O2[r] = (P2[3 * r] + P2[3 * r + 1] + P2[3 * r + 2]) * third;
        for (int c = 0; c < 3; c++) { Pr1[3 * r + c] = P1[3 * r + c] - O1[r]; Pr2[3 * r + c] = P2[3 * r + c] - O2[r]; }
    }
    float M[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += (double)Pr2[3 * i + k] * (double)Pr1[3 * j + k]; M[3 * i + j] = (float)a; }
    const double N11 = M[0] + M[4] + M[8], N12 = M[5] - M[7], N13 = M[6] - M[2], N14 = M[1] - M[3];
    const double N22 = M[0] - M[4] - M[8], N23 = M[1] + M[3], N24 = M[6] + M[2];
    const double N33 = -M[0] + M[4] - M[8], N34 = M[5] + M[7], N44 = -M[0] - M[4] + M[8];
    float Nm[16] = { (float)N11, (float)N12, (float)N13, (float)N14, (float)N12, (float)N22, (float)N23, (float)N24,
                     (float)N13, (float)N23, (float)N33, (float)N34, (float)N14, (float)N24, (float)N34, (float)N44 };
    float eval[4], evec[16];
    sgx_jacobi4(Nm, eval, evec);
    float vec[3] = { evec[1], evec[2], evec[3] };
    const double nv = sqrt((double)vec[0] * vec[0] + (double)vec[1] * vec[1] + (double)vec[2] * vec[2]);
    const double ang = atan2(nv, (double)evec[0]);
    const float sc = (float)((2 * ang) * (1. / nv));
    for (int i = 0; i < 3; i++) vec[i] = vec[i] * sc;
    float *T12 = out, *T21 = out + 16, *R = out + 32, *t = out + 41;
    sgx_rodrigues(vec, R);
    float P3[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) P3[3 * i + j] = sgx_gemm_small3(R + 3 * i, Pr2 + j, 3, 1.0, 0.0, 0.f);
    float s12 = 1.0f;
    if (!fix_scale) {
        double nom = 0;                                          // Mat::dot, four products per step
        nom += (double)Pr1[0] * P3[0] + (double)Pr1[1] * P3[1] + (double)Pr1[2] * P3[2] + (double)Pr1[3] * P3[3];
        nom += (double)Pr1[4] * P3[4] + (double)Pr1[5] * P3[5] + (double)Pr1[6] * P3[6] + (double)Pr1[7] * P3[7];
        nom += (double)Pr1[8] * P3[8];
        double den = 0;
        for (int i = 0; i < 9; i++) { const float sq = P3[i] * P3[i]; den += (double)sq; }
        s12 = (float)(nom / den);
    }
    for (int i = 0; i < 3; i++) t[i] = sgx_gemm_small3(R + 3 * i, O2, 1, -(double)s12, 1.0, O1[i]);
    for (int i = 0; i < 16; i++) { T12[i] = 0.f; T21[i] = 0.f; }
    T12[15] = 1.f; T21[15] = 1.f;
    float sRinv[9];
    const float inv_s = (float)(1.0 / (double)s12);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T12[4 * i + j] = R[3 * i + j] * s12; sRinv[3 * i + j] = R[3 * j + i] * inv_s; T21[4 * i + j] = sRinv[3 * i + j]; }
    for (int i = 0; i < 3; i++) { T12[4 * i + 3] = t[i]; T21[4 * i + 3] = sgx_gemm_small3(sRinv + 3 * i, t, 1, -1.0, 0.0, 0.f); }
    out[44] = s12;
}

// one correspondence against one hypothesis: Project + the two error tests of CheckInliers
SGX_DEV bool sgx_s3_inlier(const SgxS3Args &A, const float *hyp, int i)
{
    float p21[2] = { 0.f, 0.f }, p12[2] = { 0.f, 0.f };
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
        const float *T = dir == 0 ? hyp : hyp + 16, *X = dir == 0 ? A.X2 + 3 * (size_t)i : A.X1 + 3 * (size_t)i, *K = dir == 0 ? A.K1 : A.K2;
        const float pc0 = sgx_gemm_small3(T, X, 1, 1.0, 1.0, T[3]), pc1 = sgx_gemm_small3(T + 4, X, 1, 1.0, 1.0, T[7]), pc2 = sgx_gemm_small3(T + 8, X, 1, 1.0, 1.0, T[11]);
        const float invz = 1 / pc2, x = pc0 * invz, y = pc1 * invz;
        float *o = dir == 0 ? p21 : p12;
        o[0] = K[0] * x + K[2]; o[1] = K[1] * y + K[3];
    }
    const float d10 = A.P1im1[2 * (size_t)i] - p21[0], d11 = A.P1im1[2 * (size_t)i + 1] - p21[1], d20 = p12[0] - A.P2im2[2 * (size_t)i], d21 = p12[1] - A.P2im2[2 * (size_t)i + 1];
    const float err1 = (float)((double)d10 * d10 + (double)d11 * d11), err2 = (float)((double)d20 * d20 + (double)d21 * d21);
    return err1 < A.maxErr1[i] && err2 < A.maxErr2[i];
}

SGX_KERNEL(256) k_sim3_ransac(SgxS3Args A)
{
    SGX_LDS int counts[SGX_S3_MAXIT];
    SGX_LDS int s_best;
    // ---- hypotheses
    SGX_THREADS_BEGIN(tid)
    for (int h = tid; h < A.n_iter; h += 256) {
        counts[h] = 0;
        const int N = A.N;
        int idx[3];
        {   // RandomInt(0, size - 1) three times with the swap-and-pop of vAvailableIndices (:166-178), in closed form
            const double R1 = (double)2147483647 + 1.0;
            const int r1 = (int)(((double)A.draws[3 * h] / R1) * N);
            const int r2 = (int)(((double)A.draws[3 * h + 1] / R1) * (N - 1));
            const int r3 = (int)(((double)A.draws[3 * h + 2] / R1) * (N - 2));
            idx[0] = r1;
            const int p1 = r1, v1 = N - 1;                        // avail[r1] = back (N - 1)
            idx[1] = r2 == p1 ? v1 : r2;
            const int back2 = (N - 2) == p1 ? v1 : N - 2;         // back after the first pop
            const int p2 = r2, v2 = back2;                        // avail[r2] = back
            idx[2] = r3 == p2 ? v2 : (r3 == p1 ? v1 : r3);
        }
        float P1[9], P2[9];
        for (int i = 0; i < 3; i++) for (int r = 0; r < 3; r++) { P1[3 * r + i] = A.X1[3 * (size_t)idx[i] + r]; P2[3 * r + i] = A.X2[3 * (size_t)idx[i] + r]; }
        sgx_compute_sim3(P1, P2, A.fix_scale, A.hyp + (size_t)h * SGX_S3_HYP);
    }
    SGX_THREADS_END
    SGX_SYNC();
    // ---- CheckInliers for every hypothesis
    SGX_THREADS_BEGIN(tid)
    const int total = A.n_iter * A.N;
    for (int q = tid; q < total; q += 256) {
        const int h = q / A.N, i = q - h * A.N;
        if (sgx_s3_inlier(A, A.hyp + (size_t)h * SGX_S3_HYP, i)) sgx_atomic_add(&counts[h], 1);
    }
    SGX_THREADS_END
    SGX_SYNC();
    // ---- the sequential accept rule (:182-201)
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) {
        int best = A.best_in, best_h = -1, found = -1, run = A.n_iter;
        for (int h = 0; h < A.n_iter; h++) {
            const int c = counts[h];
            if (c >= best) { best = c; best_h = h; if (c > A.min_inliers) { found = h; run = h + 1; break; } }
        }
        A.result[0] = found; A.result[1] = run; A.result[2] = best_h; A.result[3] = best_h >= 0 ? best : 0;
        s_best = best_h;
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (s_best >= 0) for (int i = tid; i < A.N; i += 256) A.inl[i] = sgx_s3_inlier(A, A.hyp + (size_t)s_best * SGX_S3_HYP, i) ? 1 : 0;
    SGX_THREADS_END
}
