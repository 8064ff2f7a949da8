// sgx_se3.h — device restatement of g2o::SE3Quat and the small fp64 pieces shared by the pose-optimisation and
// bundle-adjustment kernels (G = src/sg-slam/Thirdparty/g2o/g2o): se3quat.h:41-296, robust_kernel_impl.cpp:78-91,
// base_edge.h:58-61, linear_solver_dense.h:105-111, Converter.cc:37-71.  Static indexing only (registers, no scratch).
#pragma once
#include "sgx_rt.h"

struct SgxSE3 { double q[4]; double t[3]; };   // quaternion x,y,z,w + translation (g2o::SE3Quat)

// a / b, correctly rounded, from r = RN(1 / b): one multiply and two fused multiply-adds (Markstein's theorem: with the correctly rounded
// reciprocal and a faithful first quotient, the corrected quotient is the correctly rounded one).  Zero / infinite / NaN operands make the
// remainder non-finite and keep the first quotient a * r, which is then already the IEEE result (+-inf, 0 or NaN); a zero remainder keeps
// it too (it is exact, and adding +0 would lose the sign of a -0 quotient).
SGX_DEV double sgx_div_by_recip(double a, double b, double r)
{
    const double q = a * r;
    const double rem = fma(-b, q, a);
    return (fabs(rem) <= 1.7976931348623157e308 && rem != 0.0) ? fma(rem, r, q) : q;
}

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
    const double rn = 1.0 / n;                    // four correctly rounded quotients q[i] / n from one division
    q[0] = sgx_div_by_recip(q[0], n, rn); q[1] = sgx_div_by_recip(q[1], n, rn); q[2] = sgx_div_by_recip(q[2], n, rn); q[3] = sgx_div_by_recip(q[3], n, rn);
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
        double st, ct; sincos(theta, &st, &ct);
        const double a = st / theta, b = (1 - ct) / (theta * theta), c = (theta - st) / (theta * theta * theta);
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

// the two values separately: the square root (and quotient) are evaluated only if some lane of the wave is beyond delta, and selected per lane
SGX_DEV double sgx_huber_rho0(double e, double delta)
{
    const double dsqr = delta * delta;
    double r = e;
    if (SGX_WAVE_ANY(!(e <= dsqr))) r = e <= dsqr ? e : 2 * sqrt(e) * delta - dsqr;
    return r;
}
SGX_DEV double sgx_huber_rho1(double e, double delta)
{
    const double dsqr = delta * delta;
    double r = 1.;
    if (SGX_WAVE_ANY(!(e <= dsqr))) r = e <= dsqr ? 1. : delta / sqrt(e);
    return r;
}

// LinearSolverDense (G/solvers/linear_solver_dense.h:105-111) factorises H with Eigen's LDLT and rejects the step
// when the factorisation is not positive (levenberg.cpp:126-127).  Here: LDL^T of the 6x6 in natural order, fully
// unrolled (registers only).  H + lambda*I is symmetric positive definite whenever the reference's pivoted LDLT
// reports "positive", and then both give the same solution to ~1e-15 relative; a non-positive or NaN pivot
// returns false (step rejected) like !isPositive().  Quotients by a pivot are products with its reciprocal (six divisions instead of 21).
SGX_DEV bool sgx_ldlt6_solve(const double Hin[6][6], const double b[6], double x[6])
{
    double L[6][6], D[6], rD[6];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = Hin[j][j];
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < j) d -= L[j][k] * L[j][k] * D[k];
        D[j] = d; rD[j] = 1.0 / d;
        if (!(d > 0)) ok = false;
#pragma unroll
        for (int i = 0; i < 6; i++) if (i > j) {
            double v = Hin[i][j];
#pragma unroll
            for (int k = 0; k < 6; k++) if (k < j) v -= L[i][k] * L[j][k] * D[k];
            L[i][j] = v * rD[j];
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
    for (int i = 0; i < 6; i++) y[i] *= rD[i];
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
// the same residual, branch-free for both edge kinds (one division: the mono quotients p/z come from the reciprocal the stereo form needs anyway)
SGX_DEV void sgx_po_residual(const SgxSE3 &T, const float *X, const float *obs, bool stereo,
                             double fx, double fy, double cx, double cy, double bf, double *err)
{
    const double Xd[3] = { (double)X[0], (double)X[1], (double)X[2] };
    double p[3]; sgx_se3_map(T, Xd, p);
    const double rz = 1.0 / p[2];
    const float invz = (float)rz;
    const double s0 = p[0] * invz * fx + cx, s1 = p[1] * invz * fy + cy, s2 = s0 - bf * invz;
    double m0 = s0, m1 = s1;
    if (SGX_WAVE_ANY(!stereo)) { m0 = sgx_div_by_recip(p[0], p[2], rz) * fx + cx; m1 = sgx_div_by_recip(p[1], p[2], rz) * fy + cy; }
    err[0] = (double)obs[0] - (stereo ? s0 : m0); err[1] = (double)obs[1] - (stereo ? s1 : m1); err[2] = stereo ? (double)obs[2] - s2 : 0.0;
}
// BaseEdge::chi2 = e . (Omega e), Omega = invSigma2 * I   (G/core/base_edge.h:58-61)
SGX_DEV double sgx_po_chi2(const double *err, double info, int stereo)
{
    double s = err[0] * (info * err[0]);
    s += err[1] * (info * err[1]);
    if (stereo) s += err[2] * (info * err[2]);
    return s;
}

