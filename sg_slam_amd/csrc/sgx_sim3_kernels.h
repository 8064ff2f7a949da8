// sgx_sim3_kernels.h — Optimizer::OptimizeSim3 on the device (tier N4 of SURVEY.md §8): src/sg-slam/src/Optimizer.cc:1046-1257 with the g2o pieces it runs
// (G = src/sg-slam/Thirdparty/g2o/g2o): Sim3 G/types/sim3.h:72-139,:230-289; VertexSim3Expmap::oplusImpl and cam_map1/2, Edge(Inverse)Sim3ProjectXYZ::computeError
// G/types/types_seven_dof_expmap.h:58-171; their linearizeOplus is commented out in the vendored g2o, so the NUMERIC Jacobian of BaseBinaryEdge applies
// (G/core/base_binary_edge.hpp:131-205: central differences, delta = 1e-9, one dimension at a time through oplus); Levenberg-Marquardt, Huber kernel and the dense LDLT as
// in k_pose_opt.  One workgroup per problem (loop closing is a rare, sequential event; a problem has a few hundred correspondences).
#pragma once
#include "sgx_rt.h"
#include "sgx_sim3.h"

struct SgxSim3Args {
    int n, fix_scale; float th2;
    float K1[4], K2[4];                                            // fx, fy, cx, cy of the two keyframes
    const float *p1c, *p2c, *obs1, *obs2, *info1, *info2;
    double *S12;                                                   // in/out: qx, qy, qz, qw, tx, ty, tz, s
    double *err;                                                   // scratch: 2 n x 2 edge errors (they persist between optimize() calls, like g2o's _error)
    uint8_t *inlier; int *iters; int *nin;
};
#define SGX_S3_NRED 36                                             /* 28 upper-triangle entries of H + 7 of b + chi2 */

// error of edge k = 2 i + (inverse ? 1 : 0) at the similarity S (or its inverse Sinv for the inverse edge)
SGX_DEV void sgx_s3_error(const SgxSim3Args &A, int k, const SgxSim3 &S, const SgxSim3 &Sinv, double e[2])
{
    const int i = k >> 1; const bool inv = k & 1;
    const float *p = (inv ? A.p1c : A.p2c) + 3 * (size_t)i, *ob = (inv ? A.obs2 : A.obs1) + 2 * (size_t)i;
    const float *K = inv ? A.K2 : A.K1;
    const double x[3] = { (double)p[0], (double)p[1], (double)p[2] };
    double m[3];
    sgx_sim3_map(inv ? Sinv : S, x, m);
    e[0] = (double)ob[0] - (m[0] / m[2] * (double)K[0] + (double)K[2]);
    e[1] = (double)ob[1] - (m[1] / m[2] * (double)K[1] + (double)K[3]);
}

SGX_KERNEL(256) k_optimize_sim3(SgxSim3Args A)
{
    SGX_LDS SgxSim3 s_est, s_inv, s_pert[28];                      // estimate, its inverse; [d] = oplus(+delta e_d), [7+d] = oplus(-delta e_d), [14+d] / [21+d] = their inverses
    SGX_LDS double part[256 * SGX_S3_NRED];
    SGX_LDS double part2[SGX_S3_NRED * 8], red[SGX_S3_NRED];
    SGX_LDS double s_x[7];
    SGX_LDS int s_ok, s_nbad;
    const int ne = 2 * A.n, NT = 256;
    const double delta = (double)sqrtf(A.th2), th2 = (double)A.th2;
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) {
        for (int i = 0; i < 4; i++) s_est.q[i] = A.S12[i];
        for (int i = 0; i < 3; i++) s_est.t[i] = A.S12[4 + i];
        s_est.s = A.S12[7]; s_nbad = 0;
    }
    for (int i = tid; i < A.n; i += NT) A.inlier[i] = 1;
    SGX_THREADS_END
    SGX_SYNC();

// sum the NRED per-thread partials in a fixed order (two levels) -> red[]
#define SGX_S3_REDUCE()                                                                                          \
    SGX_SYNC();                                                                                                  \
    SGX_THREADS_BEGIN(tid)                                                                                       \
    for (int idx = tid; idx < SGX_S3_NRED * 8; idx += NT) { const int c = idx >> 3, g = idx & 7; double s = 0; for (int l = 0; l < 32; l++) s += part[(g * 32 + l) * SGX_S3_NRED + c]; part2[c * 8 + g] = s; } \
    SGX_THREADS_END                                                                                              \
    SGX_SYNC();                                                                                                  \
    SGX_THREADS_BEGIN(tid)                                                                                       \
    if (tid < SGX_S3_NRED) { double s = 0; for (int g = 0; g < 8; g++) s += part2[tid * 8 + g]; red[tid] = s; }  \
    SGX_THREADS_END                                                                                              \
    SGX_SYNC();
// errors of all alive edges at s_est (stored: they are what chi2() reads later) and the robust chi2 -> red[35]
#define SGX_S3_ERRORS()                                                                                          \
    SGX_THREADS_BEGIN(tid) if (tid == 0) sgx_sim3_inverse(s_est, s_inv); SGX_THREADS_END                         \
    SGX_SYNC();                                                                                                  \
    SGX_THREADS_BEGIN(tid)                                                                                       \
    double chi = 0;                                                                                              \
    for (int k = tid; k < ne; k += NT) {                                                                         \
        if (!A.inlier[k >> 1]) continue;                                                                         \
        double e[2]; sgx_s3_error(A, k, s_est, s_inv, e);                                                        \
        A.err[2 * k] = e[0]; A.err[2 * k + 1] = e[1];                                                            \
        const double info = (double)((k & 1) ? A.info2 : A.info1)[k >> 1];                                       \
        double r0, r1; sgx_huber(e[0] * (info * e[0]) + e[1] * (info * e[1]), delta, &r0, &r1); chi += r0;       \
    }                                                                                                            \
    for (int c = 0; c < SGX_S3_NRED; c++) part[tid * SGX_S3_NRED + c] = c == 35 ? chi : 0.0;                     \
    SGX_THREADS_END                                                                                              \
    SGX_S3_REDUCE()

    for (int call = 0; call < 2; call++) {                        // optimizer.optimize(5), then optimize(nMoreIterations) on the survivors
        int iterations = 5;
        if (call == 1) {
            // ---- check inliers (Optimizer.cc:1186-1204): chi2() of the stored errors
            SGX_THREADS_BEGIN(tid)
            int nb = 0;
            for (int i = tid; i < A.n; i += NT) {
                const double i1 = (double)A.info1[i], i2 = (double)A.info2[i];
                const double c12 = A.err[4 * i] * (i1 * A.err[4 * i]) + A.err[4 * i + 1] * (i1 * A.err[4 * i + 1]);
                const double c21 = A.err[4 * i + 2] * (i2 * A.err[4 * i + 2]) + A.err[4 * i + 3] * (i2 * A.err[4 * i + 3]);
                if (c12 > th2 || c21 > th2) { A.inlier[i] = 0; nb++; }
            }
            if (nb) sgx_atomic_add(&s_nbad, nb);
            SGX_THREADS_END
            SGX_SYNC();
            iterations = s_nbad > 0 ? 10 : 5;
            if (A.n - s_nbad < 10) {                                // return 0 before the estimate is read back (:1212-1213)
                SGX_THREADS_BEGIN(tid) if (tid == 0) { *A.nin = 0; A.iters[1] = 0; } SGX_THREADS_END
                return;
            }
        }
        double lambda = -1, ni = 2; int nBadLM = 0, iters = 0;
        for (int it = 0; it < iterations; it++) {
            SGX_S3_ERRORS()
            double currentChi = red[35], tempChi = currentChi; const double iniChi = currentChi;
            // ---- buildSystem with numeric Jacobians: the 14 perturbed estimates (and their inverses) are shared by all edges
            SGX_THREADS_BEGIN(tid)
            if (tid < 14) {
                const int d = tid % 7; double add[7] = { 0, 0, 0, 0, 0, 0, 0 };
                add[d] = tid < 7 ? 1e-9 : -1e-9;
                SgxSim3 p; sgx_sim3_oplus(s_est, add, A.fix_scale, p);
                SgxSim3 pi; sgx_sim3_inverse(p, pi);
                s_pert[tid] = p; s_pert[14 + tid] = pi;
            }
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            double acc[SGX_S3_NRED];
            for (int c = 0; c < SGX_S3_NRED; c++) acc[c] = 0;
            const double scalar = 1.0 / (2 * 1e-9);
            for (int k = tid; k < ne; k += NT) {
                if (!A.inlier[k >> 1]) continue;
                double J[2][7];
                for (int d = 0; d < 7; d++) {
                    double ep[2], em[2];
                    sgx_s3_error(A, k, s_pert[d], s_pert[14 + d], ep); sgx_s3_error(A, k, s_pert[7 + d], s_pert[21 + d], em);
                    J[0][d] = scalar * (ep[0] - em[0]); J[1][d] = scalar * (ep[1] - em[1]);
                }
                const double info = (double)((k & 1) ? A.info2 : A.info1)[k >> 1], e0 = A.err[2 * k], e1 = A.err[2 * k + 1];
                double r0, rho1; sgx_huber(e0 * (info * e0) + e1 * (info * e1), delta, &r0, &rho1);
                const double w = rho1 * info;
                int c = 0;
                for (int a = 0; a < 7; a++) {
                    acc[28 + a] -= rho1 * (J[0][a] * (info * e0) + J[1][a] * (info * e1));
                    for (int b = a; b < 7; b++) acc[c++] += J[0][a] * w * J[0][b] + J[1][a] * w * J[1][b];
                }
            }
            for (int c = 0; c < SGX_S3_NRED; c++) part[tid * SGX_S3_NRED + c] = acc[c];
            SGX_THREADS_END
            SGX_S3_REDUCE()
            double H[7][7], b[7];
            { int c = 0; for (int a = 0; a < 7; a++) { b[a] = red[28 + a]; for (int q = a; q < 7; q++) { H[a][q] = red[c]; H[q][a] = red[c]; c++; } } }
            if (it == 0) { double maxd = 0; for (int j = 0; j < 7; j++) if (fabs(H[j][j]) > maxd) maxd = fabs(H[j][j]); lambda = 1e-5 * maxd; ni = 2; nBadLM = 0; }
            double rho = 0; int qmax = 0;
            const SgxSim3 base = s_est;                              // push: every trial of this iteration starts from here until one is accepted
            SgxSim3 cur = base;
            do {
                SGX_SYNC();
                SGX_THREADS_BEGIN(tid)
                if (tid == 0) {
                    double Hl[7][7], x[7] = { 0, 0, 0, 0, 0, 0, 0 };
                    for (int a = 0; a < 7; a++) for (int q = 0; q < 7; q++) Hl[a][q] = H[a][q] + (a == q ? lambda : 0.0);
                    s_ok = sgx_ldlt7_solve(Hl, b, x) ? 1 : 0;
                    for (int a = 0; a < 7; a++) s_x[a] = x[a];
                    SgxSim3 upd; sgx_sim3_oplus(cur, x, A.fix_scale, upd); s_est = upd;
                }
                SGX_THREADS_END
                SGX_SYNC();
                SGX_S3_ERRORS()
                tempChi = s_ok ? red[35] : DBL_MAX;
                rho = currentChi - tempChi;
                double scale = 0; for (int j = 0; j < 7; j++) scale += s_x[j] * (lambda * s_x[j] + b[j]);
                scale += 1e-3; rho /= scale;
                if (rho > 0 && tempChi <= DBL_MAX && tempChi == tempChi && s_ok) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                    const double sf = alpha > 1. / 3. ? alpha : 1. / 3.;
                    lambda *= sf; ni = 2; currentChi = tempChi; cur = s_est;                  // discardTop
                } else {
                    lambda *= ni; ni *= 2;                                                   // pop: the stored edge errors stay those of the rejected trial
                    SGX_SYNC();
                    SGX_THREADS_BEGIN(tid) if (tid == 0) s_est = cur; SGX_THREADS_END
                    SGX_SYNC();
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            iters = it + 1;
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid) if (tid == 0) A.iters[call] = iters; SGX_THREADS_END
        SGX_SYNC();
    }
    // ---- final inlier count on the stored errors (:1222-1238), recover the estimate
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) s_nbad = 0;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    int nin = 0;
    for (int i = tid; i < A.n; i += NT) {
        if (!A.inlier[i]) continue;
        const double i1 = (double)A.info1[i], i2 = (double)A.info2[i];
        const double c12 = A.err[4 * i] * (i1 * A.err[4 * i]) + A.err[4 * i + 1] * (i1 * A.err[4 * i + 1]);
        const double c21 = A.err[4 * i + 2] * (i2 * A.err[4 * i + 2]) + A.err[4 * i + 3] * (i2 * A.err[4 * i + 3]);
        if (c12 > th2 || c21 > th2) A.inlier[i] = 0; else nin++;
    }
    if (nin) sgx_atomic_add(&s_nbad, nin);
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) {
        *A.nin = s_nbad;
        for (int i = 0; i < 4; i++) A.S12[i] = s_est.q[i];
        for (int i = 0; i < 3; i++) A.S12[4 + i] = s_est.t[i];
        A.S12[7] = s_est.s;
    }
    SGX_THREADS_END
#undef SGX_S3_ERRORS
#undef SGX_S3_REDUCE
}
