// sgx_poseopt_kernels.h — HIP kernel for Optimizer::PoseOptimization (fp64 Levenberg-Marquardt
// over one SE3 pose and N unary reprojection edges), one 64-lane wave per frame.  Phase style.
// Reference behaviour: src/sg-slam/src/Optimizer.cc:239-451 and the vendored g2o it drives
// (G = src/sg-slam/Thirdparty/g2o/g2o; cited inline; the CPU restatement is oracle/poseopt_oracle.c).
#pragma once
#include "sgx_rt.h"
#include "sgx_types.h"
#include "sgx_block.h"

#define SGX_PO_CAP 1280
#define SGX_PO_THREADS 256
#define SGX_PO_NRED 28            /* 21 upper-triangular H entries + 6 b entries + chi */

struct SgxSE3 { double q[4]; double t[3]; };   // quaternion x,y,z,w + translation (g2o::SE3Quat)

SGX_DEV void sgx_quat_from_R(const double R[3][3], double q[4])
{   // Eigen Quaterniond(Matrix3d) (Shepperd branches); written with static indices only (no scratch)
    double t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[2][1] - R[1][2]) * t; q[1] = (R[0][2] - R[2][0]) * t; q[2] = (R[1][0] - R[0][1]) * t;
    } else {
        int i = 0; if (R[1][1] > R[0][0]) i = 1;
        const double rii = i == 0 ? R[0][0] : R[1][1];
        if (R[2][2] > rii) i = 2;
        if (i == 0) {            // j = 1, k = 2
            t = sqrt(R[0][0] - R[1][1] - R[2][2] + 1.0); q[0] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[2][1] - R[1][2]) * t; q[1] = (R[1][0] + R[0][1]) * t; q[2] = (R[2][0] + R[0][2]) * t;
        } else if (i == 1) {     // j = 2, k = 0
            t = sqrt(R[1][1] - R[2][2] - R[0][0] + 1.0); q[1] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[0][2] - R[2][0]) * t; q[2] = (R[2][1] + R[1][2]) * t; q[0] = (R[0][1] + R[1][0]) * t;
        } else {                 // j = 0, k = 1
            t = sqrt(R[2][2] - R[0][0] - R[1][1] + 1.0); q[2] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[1][0] - R[0][1]) * t; q[0] = (R[0][2] + R[2][0]) * t; q[1] = (R[1][2] + R[2][1]) * t;
        }
    }
}
SGX_DEV void sgx_quat_normalize_rot(double q[4])
{   // SE3Quat::normalizeRotation, G/types/se3quat.h:280-285
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
SGX_DEV void sgx_quat_rotate(const double q[4], const double v[3], double o[3])
{   // Eigen QuaternionBase::_transformVector
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
SGX_DEV void sgx_quat_mul(const double a[4], const double b[4], double o[4])
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
SGX_DEV void sgx_se3_map(const SgxSE3 &T, const double x[3], double o[3])
{ sgx_quat_rotate(T.q, x, o); o[0] += T.t[0]; o[1] += T.t[1]; o[2] += T.t[2]; }

SGX_DEV void sgx_se3_exp(const double u[6], SgxSE3 &out)
{   // SE3Quat::exp, G/types/se3quat.h:223-257 (omega = u[0..2], upsilon = u[3..5]), incl. the small-angle branch
    const double w0 = u[0], w1 = u[1], w2 = u[2];
    const double theta = sqrt(w0 * w0 + w1 * w1 + w2 * w2);
    const double O[3][3] = { { 0, -w2, w1 }, { w2, 0, -w0 }, { -w1, w0, 0 } };
    double O2[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
    }
    double R[3][3], V[3][3];
    if (theta < 0.00001) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) { R[i][j] = (i == j) + O[i][j] + O2[i][j]; V[i][j] = R[i][j]; }
        }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                R[i][j] = (i == j) + a * O[i][j] + b * O2[i][j];
                V[i][j] = (i == j) + b * O[i][j] + c * O2[i][j];
            }
        }
    }
    sgx_quat_from_R(R, out.q);
#pragma unroll
    for (int i = 0; i < 3; i++) out.t[i] = V[i][0] * u[3] + V[i][1] * u[4] + V[i][2] * u[5];
    sgx_quat_normalize_rot(out.q);
}
SGX_DEV void sgx_se3_mul(const SgxSE3 &a, const SgxSE3 &b, SgxSE3 &o)
{   // SE3Quat::operator*, G/types/se3quat.h:104-110
    double rt[3]; sgx_quat_rotate(a.q, b.t, rt);
    SgxSE3 r;
    r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
    sgx_quat_mul(a.q, b.q, r.q); sgx_quat_normalize_rot(r.q);
    o = r;
}
SGX_DEV void sgx_se3_from_cv(const float *T, SgxSE3 &o)
{   // Converter::toSE3Quat, src/sg-slam/src/Converter.cc:37-47
    double R[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) R[i][j] = (double)T[4 * i + j];
        o.t[i] = (double)T[4 * i + 3];
    }
    sgx_quat_from_R(R, o.q); sgx_quat_normalize_rot(o.q);
}
SGX_DEV void sgx_se3_to_cv(const SgxSE3 &s, float *T)
{   // Converter::toCvMat(SE3Quat), Converter.cc:49-71 (Eigen toRotationMatrix)
    const double *q = s.q;
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    T[0] = (float)(1 - (tyy + tzz)); T[1] = (float)(txy - twz); T[2] = (float)(txz + twy); T[3] = (float)s.t[0];
    T[4] = (float)(txy + twz); T[5] = (float)(1 - (txx + tzz)); T[6] = (float)(tyz - twx); T[7] = (float)s.t[1];
    T[8] = (float)(txz - twy); T[9] = (float)(tyz + twx); T[10] = (float)(1 - (txx + tyy)); T[11] = (float)s.t[2];
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// RobustKernelHuber::robustify, G/core/robust_kernel_impl.cpp:78-91 (rho0, rho1)
SGX_DEV void sgx_huber(double e, double delta, double *rho0, double *rho1)
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { *rho0 = e; *rho1 = 1.; }
    else { const double sq = sqrt(e); *rho0 = 2 * sq * delta - dsqr; *rho1 = delta / sq; }
}

// LinearSolverDense (G/solvers/linear_solver_dense.h:105-111) factorises H with Eigen's LDLT and rejects the step
// when the factorisation is not positive (levenberg.cpp:126-127).  Here: LDL^T of the 6x6 in natural order, fully
// unrolled (registers only).  H + lambda*I is symmetric positive definite whenever the reference's pivoted LDLT
// reports "positive", and then both give the same solution to ~1e-15 relative; a non-positive or NaN pivot
// returns false (step rejected) like !isPositive().
SGX_DEV bool sgx_ldlt6_solve(const double Hin[6][6], const double b[6], double x[6])
{
    double L[6][6], D[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = Hin[j][j];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < j) d -= L[j][k] * L[j][k] * D[k];
        D[j] = d;
        if (!(d > 0)) ok = false;
#pragma unroll
        for (int i = 0; i < 6; i++) if (i > j) {
            double v = Hin[i][j];
#pragma unroll
            for (int k = 0; k < 6; k++) if (k < j) v -= L[i][k] * L[j][k] * D[k];
            L[i][j] = v / d;
        }
    }
    if (!ok) return false;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < i) v -= L[i][k] * y[k];
        y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] /= D[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = y[i];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k > i) v -= L[k][i] * x[k];
        x[i] = v;
    }
    return true;
}

// Edge(Stereo)SE3ProjectXYZOnlyPose::computeError, G/types/types_six_dof_expmap.h:153-157,184-188; projections .cpp:290-306
// (mono: project2d then *f + c; stereo: invz is a FLOAT in the reference)
SGX_DEV void sgx_po_edge_error(const SgxSE3 &T, const float *X, const float *obs, int stereo,
                               double fx, double fy, double cx, double cy, double bf, double *err)
{
    const double Xd[3] = { (double)X[0], (double)X[1], (double)X[2] };
    double p[3]; sgx_se3_map(T, Xd, p);
    if (!stereo) {
        const double px = p[0] / p[2], py = p[1] / p[2];
        err[0] = (double)obs[0] - (px * fx + cx); err[1] = (double)obs[1] - (py * fy + cy); err[2] = 0;
    } else {
        const float invz = (float)(1.0 / p[2]);          // 1.0f/double -> double division, rounded to float
        const double r0 = p[0] * invz * fx + cx, r1 = p[1] * invz * fy + cy, r2 = r0 - bf * invz;
        err[0] = (double)obs[0] - r0; err[1] = (double)obs[1] - r1; err[2] = (double)obs[2] - r2;
    }
}
// BaseEdge::chi2 = e . (Omega e), Omega = invSigma2 * I   (G/core/base_edge.h:58-61)
SGX_DEV double sgx_po_chi2(const double *err, double info, int stereo)
{
    double s = err[0] * (info * err[0]);
    s += err[1] * (info * err[1]);
    if (stereo) s += err[2] * (info * err[2]);
    return s;
}

// ---------------------------------------------------------------------------------------------
// k_pose_opt: one 256-thread workgroup per frame.  Edge e <-> keypoint i with a map point (ascending i, as the
// reference inserts them, Optimizer.cc:280-360).  Threads own edges e = tid, tid+256, ...; the
// 6x6 system, the chi2 sums and the LM control flow are wave-uniform (each lane evaluates the same
// scalar code on values reduced through LDS), so the kernel follows the reference's accept/reject,
// lambda schedule and stop rules statement for statement (levenberg.cpp:61-164).
// Sums over edges are reduced in a fixed order (thread-strided partials, 8 groups of 32, then 8):
// deterministic, and within ~1e-15 relative of the reference's sequential order.
// mp_index (optional): map point of keypoint i is table row mp_index[i] (-1 = none); otherwise has_mp/xw are per keypoint.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(SGX_PO_THREADS) k_pose_opt(int cap, const uint8_t *keys_raw, const float *uright, const int *n_kp,
                                      const int *mp_index, const uint8_t *has_mp, const float *xw, int xw_pitch,
                                      SgxScales inv_sigma2, SgxCam cam, float *Tcw, uint8_t *outlier, int *n_inliers)
{
    SGX_LDS float e_obs[SGX_PO_CAP * 3], e_xw[SGX_PO_CAP * 3], e_info[SGX_PO_CAP];
    SGX_LDS double e_err[SGX_PO_CAP * 3];
    SGX_LDS uint16_t e_kp[SGX_PO_CAP];
    SGX_LDS uint8_t e_flags[SGX_PO_CAP];          // bit0 stereo, bit1 level==1 (excluded), bit2 robust kernel on, bit3 outlier flag
    SGX_LDS double part[SGX_PO_THREADS * SGX_PO_NRED];
    SGX_LDS double part2[SGX_PO_NRED * 8];
    SGX_LDS double red[SGX_PO_NRED];
    SGX_LDS int scan[SGX_PO_THREADS];
    SGX_LDS int s_ne, s_nbad;

    const int f = (int)blockIdx.x;
    const int N = min(n_kp[f], cap);
    const int NT = SGX_PO_THREADS;
    const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);   // Optimizer.cc:272-273 (float)
    const double fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy, bf = cam.bf;

    // ---- build the edge list in ascending keypoint order: per-thread contiguous chunks + block scan
    const int CH = (N + NT - 1) / NT;
    SGX_THREADS_BEGIN(tid)
    int c = 0;
    for (int i = tid * CH; i < min(N, (tid + 1) * CH); i++) {
        const size_t o = (size_t)f * cap + i;
        c += mp_index ? (mp_index[o] >= 0) : (has_mp[o] != 0);
    }
    scan[tid] = c;
    for (int i = tid; i < cap; i += NT) outlier[(size_t)f * cap + i] = 0;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    sgx_block_exclusive_scan_i32(scan, NT, &s_ne, tid);
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    int ne_ = scan[tid];
    for (int i = tid * CH; i < min(N, (tid + 1) * CH); i++) {
        const size_t o = (size_t)f * cap + i;
        int src = -1;
        if (mp_index) { const int m = mp_index[o]; if (m >= 0) src = m; }
        else if (has_mp[o]) src = i;
        if (src < 0) continue;
        const float *kp = (const float *)(keys_raw + o * 28);
        const float ur = uright[o];
        e_obs[3 * ne_] = kp[0]; e_obs[3 * ne_ + 1] = kp[1]; e_obs[3 * ne_ + 2] = ur;
        const float *X = xw + ((size_t)f * xw_pitch + src) * 3;
        e_xw[3 * ne_] = X[0]; e_xw[3 * ne_ + 1] = X[1]; e_xw[3 * ne_ + 2] = X[2];
        e_info[ne_] = inv_sigma2.s[((const int *)kp)[5]];
        e_kp[ne_] = (uint16_t)i;
        e_flags[ne_] = (uint8_t)((ur < 0 ? 0 : 1) | 4);        // mono iff mvuRight<0 (Optimizer.cc:286); Huber on
        ne_++;
    }
    SGX_THREADS_END
    SGX_SYNC();
    const int ne = s_ne;
    if (ne < 3) {                                                   // Optimizer.cc:364-365
        SGX_THREADS_BEGIN(tid) if (tid == 0) n_inliers[f] = 0; SGX_THREADS_END
        return;
    }

    SgxSE3 est;
    float T0[16];
    for (int i = 0; i < 16; i++) T0[i] = Tcw[16 * f + i];
    int nBad = 0;

// evaluates errors of the active (level-0) edges at `est`, accumulates the robust chi2 into part[lane][27]
#define SGX_PO_ERRORS()                                                                                     \
    SGX_THREADS_BEGIN(tid)                                                                                  \
    double chi = 0;                                                                                         \
    for (int e = tid; e < ne; e += NT) {                                                                    \
        const int fl = e_flags[e];                                                                          \
        if (fl & 2) continue;                                                                               \
        sgx_po_edge_error(est, e_xw + 3 * e, e_obs + 3 * e, fl & 1, fx, fy, cx, cy, bf, e_err + 3 * e);      \
        const double c2 = sgx_po_chi2(e_err + 3 * e, (double)e_info[e], fl & 1);                            \
        if (fl & 4) { double r0, r1; sgx_huber(c2, (fl & 1) ? deltaStereo : deltaMono, &r0, &r1); chi += r0; } \
        else chi += c2;                                                                                     \
    }                                                                                                       \
    part[tid * SGX_PO_NRED + 27] = chi;                                                                     \
    SGX_THREADS_END                                                                                         \
    SGX_SYNC();                                                                                             \
    SGX_THREADS_BEGIN(tid)                                                                                  \
    if (tid < 8) { double s = 0; for (int l = 0; l < 32; l++) s += part[(tid * 32 + l) * SGX_PO_NRED + 27]; part2[27 * 8 + tid] = s; } \
    SGX_THREADS_END                                                                                         \
    SGX_SYNC();                                                                                             \
    SGX_THREADS_BEGIN(tid)                                                                                  \
    if (tid == 0) { double s = 0; for (int l = 0; l < 8; l++) s += part2[27 * 8 + l]; red[27] = s; }        \
    SGX_THREADS_END                                                                                         \
    SGX_SYNC();

    for (int round = 0; round < 4; round++) {
        sgx_se3_from_cv(T0, est);                                   // Optimizer.cc:377: every round restarts from pFrame->mTcw
        double lambda = -1, ni = 2; int nBadLM = 0;
        bool fresh = false; double freshChi = 0;        // e_err / chi already evaluated at `est` by an accepted trial
        for (int it = 0; it < 10; it++) {
            if (!fresh) { SGX_PO_ERRORS() freshChi = red[27]; }    // computeActiveErrors + activeRobustChi2 (levenberg.cpp:73-80)
            double currentChi = freshChi;
            double tempChi = currentChi;
            const double iniChi = currentChi;
            // ---- buildSystem: b -= rho1 * J^T (Omega e), H += J^T (rho1 Omega) J   (base_unary_edge.hpp:43-72)
            SGX_THREADS_BEGIN(tid)
            double acc[27];
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] = 0;
            for (int e = tid; e < ne; e += NT) {
                const int fl = e_flags[e];
                if (fl & 2) continue;
                const int stereo = fl & 1;
                const double Xd[3] = { (double)e_xw[3 * e], (double)e_xw[3 * e + 1], (double)e_xw[3 * e + 2] };
                double p[3]; sgx_se3_map(est, Xd, p);
                const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
                double J[3][6];                                      // types_six_dof_expmap.cpp:266-288, 335-364
                J[0][0] = x * y * invz_2 * fx; J[0][1] = -(1 + (x * x * invz_2)) * fx; J[0][2] = y * invz * fx;
                J[0][3] = -invz * fx; J[0][4] = 0; J[0][5] = x * invz_2 * fx;
                J[1][0] = (1 + y * y * invz_2) * fy; J[1][1] = -x * y * invz_2 * fy; J[1][2] = -x * invz * fy;
                J[1][3] = 0; J[1][4] = -invz * fy; J[1][5] = y * invz_2 * fy;
                J[2][0] = J[0][0] - bf * y * invz_2; J[2][1] = J[0][1] + bf * x * invz_2; J[2][2] = J[0][2];
                J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - bf * invz_2;
                const double info = (double)e_info[e];
                const double er[3] = { e_err[3 * e], e_err[3 * e + 1], stereo ? e_err[3 * e + 2] : 0.0 };
                if (!stereo) {            // mono edge: the third row does not exist (adds exact zeros below)
#pragma unroll
                    for (int a = 0; a < 6; a++) J[2][a] = 0;
                }
                double rho1 = 1.0;
                if (fl & 4) { double r0; sgx_huber(sgx_po_chi2(er, info, stereo), stereo ? deltaStereo : deltaMono, &r0, &rho1); }
                const double w = rho1 * info;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    const double s = J[0][a] * (info * er[0]) + J[1][a] * (info * er[1]) + J[2][a] * (info * er[2]);
                    acc[21 + a] -= rho1 * s;
#pragma unroll
                    for (int c = 0; c < 6; c++) if (c >= a) {
                        const double h = J[0][a] * w * J[0][c] + J[1][a] * w * J[1][c] + J[2][a] * w * J[2][c];
                        acc[a * 6 - (a * (a - 1)) / 2 + (c - a)] += h;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 27; k++) part[tid * SGX_PO_NRED + k] = acc[k];
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            if (tid < 27 * 8) { const int c = tid >> 3, j = tid & 7; double s = 0; for (int l = 0; l < 32; l++) s += part[(j * 32 + l) * SGX_PO_NRED + c]; part2[c * 8 + j] = s; }
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            if (tid < 27) { double s = 0; for (int l = 0; l < 8; l++) s += part2[tid * 8 + l]; red[tid] = s; }
            SGX_THREADS_END
            SGX_SYNC();
            double H[6][6], b[6];
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int c = 0; c < 6; c++) if (c >= a) { const double v = red[a * 6 - (a * (a - 1)) / 2 + (c - a)]; H[a][c] = v; H[c][a] = v; }
                b[a] = red[21 + a];
            }
            if (it == 0) {                                           // computeLambdaInit, levenberg.cpp:166-180 (tau = 1e-5)
                double maxd = 0;
#pragma unroll
                for (int j = 0; j < 6; j++) if (fabs(H[j][j]) > maxd) maxd = fabs(H[j][j]);
                lambda = 1e-5 * maxd; ni = 2; nBadLM = 0;
            }
            double rho = 0; int qmax = 0;
            do {
                const SgxSE3 backup = est;                           // push
                double Hl[6][6];
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int c = 0; c < 6; c++) Hl[a][c] = H[a][c] + (a == c ? lambda : 0.0);
                }
                double x[6] = { 0, 0, 0, 0, 0, 0 };
                const bool ok2 = sgx_ldlt6_solve(Hl, b, x);
                SgxSE3 ex; sgx_se3_exp(x, ex);
                SgxSE3 upd; sgx_se3_mul(ex, est, upd); est = upd;     // VertexSE3Expmap::oplusImpl
                SGX_PO_ERRORS()
                tempChi = red[27];
                if (!ok2) tempChi = 1.7976931348623157e308;
                rho = currentChi - tempChi;
                double scale = 0;
#pragma unroll
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    const double r21 = 2 * rho - 1;
                    double alpha = 1. - r21 * r21 * r21;
                    alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                    const double sf = alpha > 1. / 3. ? alpha : 1. / 3.;
                    lambda *= sf; ni = 2; currentChi = tempChi; fresh = true; freshChi = tempChi;
                } else { lambda *= ni; ni *= 2; est = backup; fresh = false; }   // pop: the edges keep the rejected trial's errors
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) break;                       // Terminate
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        // ---- classify edges (Optimizer.cc:383-438)
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) s_nbad = 0;
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        int bad = 0;
        for (int e = tid; e < ne; e += NT) {
            int fl = e_flags[e];
            const int stereo = fl & 1;
            if (fl & 8) sgx_po_edge_error(est, e_xw + 3 * e, e_obs + 3 * e, stereo, fx, fy, cx, cy, bf, e_err + 3 * e);
            const float chi2 = (float)sgx_po_chi2(e_err + 3 * e, (double)e_info[e], stereo);
            if (chi2 > (stereo ? 7.815f : 5.991f)) { fl |= (8 | 2); bad++; } else { fl &= ~(8 | 2); }
            if (round == 2) fl &= ~4;
            e_flags[e] = (uint8_t)fl;
        }
        if (bad) sgx_atomic_add(&s_nbad, bad);
        SGX_THREADS_END
        SGX_SYNC();
        nBad = s_nbad;
        if (ne < 10) break;                                          // optimizer.edges().size()<10
    }
#undef SGX_PO_ERRORS
    SGX_THREADS_BEGIN(tid)
    for (int e = tid; e < ne; e += NT) outlier[(size_t)f * cap + e_kp[e]] = (e_flags[e] & 8) ? 1 : 0;
    if (tid == 0) { float To[16]; sgx_se3_to_cv(est, To); for (int i = 0; i < 16; i++) Tcw[16 * f + i] = To[i]; n_inliers[f] = ne - nBad; }
    SGX_THREADS_END
}
