// sgx_sim3_kernels.h — Optimizer::OptimizeSim3 on the device (tier N4 of SURVEY.md §8): src/sg-slam/src/Optimizer.cc:1046-1257 with the g2o pieces it runs
// (G = src/sg-slam/Thirdparty/g2o/g2o): Sim3 G/types/sim3.h:72-139,:230-289; VertexSim3Expmap::oplusImpl and cam_map1/2, Edge(Inverse)Sim3ProjectXYZ::computeError
// G/types/types_seven_dof_expmap.h:58-171; their linearizeOplus is commented out in the vendored g2o, so the NUMERIC Jacobian of BaseBinaryEdge applies
// (G/core/base_binary_edge.hpp:131-205: central differences, delta = 1e-9, one dimension at a time through oplus); Levenberg-Marquardt, Huber kernel and the dense LDLT as
// in k_pose_opt.  One workgroup per problem (loop closing is a rare, sequential event; a problem has a few hundred correspondences).
#pragma once
#include "sgx_rt.h"
#include "sgx_se3.h"
#include <float.h>
#include <math.h>

struct SgxSim3 { double q[4]; double t[3]; double s; };          // quaternion x,y,z,w (NOT normalised: g2o::Sim3 never normalises), translation, scale

SGX_DEV void sgx_mat3_mul(const double A[3][3], const double B[3][3], double C[3][3])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}
SGX_DEV void sgx_sim3_exp(const double u[7], SgxSim3 &o)
{   // Sim3(const Vector7d &update), sim3.h:72-139
    const double w0 = u[0], w1 = u[1], w2 = u[2], sigma = u[6];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double O[3][3] = { { 0, -w2, w1 }, { w2, 0, -w0 }, { -w1, w0, 0 } };
    double O2[3][3], R[3][3];
    sgx_mat3_mul(O, O, O2);
    o.s = exp(sigma);
    const double eps = 0.00001;
    double A, B, C, ra = 1.0, rb = 1.0;                            // R = (I + ra O) + rb O2
    if (fabs(sigma) < eps) {
        C = 1;
        if (theta < eps) { A = 1. / 2.; B = 1. / 6.; }
        else { const double theta2 = theta * theta; A = (1 - cos(theta)) / theta2; B = (theta - sin(theta)) / (theta2 * theta); ra = sin(theta) / theta; rb = (1 - cos(theta)) / (theta * theta); }
    } else {
        C = (o.s - 1) / sigma;
        if (theta < eps) { const double sigma2 = sigma * sigma; A = ((sigma - 1) * o.s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * o.s) / (sigma2 * sigma); }
        else {
            ra = sin(theta) / theta; rb = (1 - cos(theta)) / (theta * theta);
            const double a = o.s * sin(theta), b = o.s * cos(theta), theta2 = theta * theta, sigma2 = sigma * sigma, c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    const bool small = theta < eps;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) R[i][j] = small ? (((i == j) + O[i][j]) + O2[i][j]) : (((i == j) + ra * O[i][j]) + rb * O2[i][j]);
    sgx_quat_from_R(R, o.q);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) s += ((A * O[i][k] + B * O2[i][k]) + C * (i == k)) * u[3 + k];
        o.t[i] = s;
    }
}
SGX_DEV void sgx_sim3_map(const SgxSim3 &S, const double x[3], double o[3]) { double r[3]; sgx_quat_rotate(S.q, x, r); o[0] = S.s * r[0] + S.t[0]; o[1] = S.s * r[1] + S.t[1]; o[2] = S.s * r[2] + S.t[2]; }
SGX_DEV void sgx_sim3_mul(const SgxSim3 &a, const SgxSim3 &b, SgxSim3 &o)
{
    SgxSim3 r; sgx_quat_mul(a.q, b.q, r.q);
    double rt[3]; sgx_quat_rotate(a.q, b.t, rt);
    r.t[0] = a.s * rt[0] + a.t[0]; r.t[1] = a.s * rt[1] + a.t[1]; r.t[2] = a.s * rt[2] + a.t[2];
    r.s = a.s * b.s; o = r;
}
SGX_DEV void sgx_sim3_inverse(const SgxSim3 &a, SgxSim3 &o)
{
    SgxSim3 r; r.q[0] = -a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = a.q[3];
    const double f = -1. / a.s; const double v[3] = { f * a.t[0], f * a.t[1], f * a.t[2] };
    sgx_quat_rotate(r.q, v, r.t); r.s = 1. / a.s; o = r;
}
SGX_DEV void sgx_sim3_oplus(const SgxSim3 &est, const double upd[7], int fix_scale, SgxSim3 &o)
{
    double u[7];
#pragma unroll
    for (int i = 0; i < 7; i++) u[i] = upd[i];
    if (fix_scale) u[6] = 0;
    SgxSim3 d; sgx_sim3_exp(u, d); sgx_sim3_mul(d, est, o);
}

// LDL^T of the damped 7x7 in natural order (see sgx_ldlt6_solve: same solution as the reference's pivoted Eigen LDLT whenever that one reports "positive")
SGX_DEV bool sgx_ldlt7_solve(const double Hin[7][7], const double b[7], double x[7])
{
    double L[7][7], D[7], rD[7];
    bool ok = true;
    for (int j = 0; j < 7; j++) {
        double d = Hin[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        D[j] = d; rD[j] = 1.0 / d;
        if (!(d > 0)) ok = false;
        for (int i = j + 1; i < 7; i++) {
            double v = Hin[i][j];
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k] * D[k];
            L[i][j] = v * rD[j];
        }
    }
    if (!ok) return false;
    double y[7];
    for (int i = 0; i < 7; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[i][k] * y[k]; y[i] = v; }
    for (int i = 0; i < 7; i++) y[i] *= rD[i];
    for (int i = 6; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 7; k++) v -= L[k][i] * x[k]; x[i] = v; }
    return true;
}

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
