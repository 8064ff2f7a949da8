// sgx_sim3.h — device restatement of g2o::Sim3 (G/types/sim3.h) and the small pieces shared by the Sim3 optimisers (k_optimize_sim3, the essential-graph kernels).
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


SGX_DEV void sgx_quat_to_R(const double q[4], double R[3][3])
{   // Eigen QuaternionBase::toRotationMatrix
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
    R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}
SGX_DEV void sgx_lu3_solve(const double Ain[3][3], const double b[3], double x[3])
{   // Eigen PartialPivLU of a 3x3 and solve (W.lu().solve(t), sim3.h:211); written out for k = 0, 1 with static indices
    double A[3][3]; double bb[3] = { b[0], b[1], b[2] };
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[i][j] = Ain[i][j];
#define SGX_SWAP_ROWS(r, s) do { _Pragma("unroll") for (int j_ = 0; j_ < 3; j_++) { const double t_ = A[r][j_]; A[r][j_] = A[s][j_]; A[s][j_] = t_; } const double u_ = bb[r]; bb[r] = bb[s]; bb[s] = u_; } while (0)
    {   // column 0: the first row of maximal magnitude
        const double a0 = fabs(A[0][0]), a1 = fabs(A[1][0]), a2 = fabs(A[2][0]);
        if (a1 > a0 && !(a2 > a1)) SGX_SWAP_ROWS(0, 1); else if (a2 > a0 && a2 > a1) SGX_SWAP_ROWS(0, 2);
    }
    A[1][0] /= A[0][0]; A[1][1] -= A[1][0] * A[0][1]; A[1][2] -= A[1][0] * A[0][2];
    A[2][0] /= A[0][0]; A[2][1] -= A[2][0] * A[0][1]; A[2][2] -= A[2][0] * A[0][2];
    if (fabs(A[2][1]) > fabs(A[1][1])) SGX_SWAP_ROWS(1, 2);
#undef SGX_SWAP_ROWS
    A[2][1] /= A[1][1]; A[2][2] -= A[2][1] * A[1][2];
    const double y0 = bb[0], y1 = bb[1] - A[1][0] * y0, y2 = bb[2] - A[2][0] * y0 - A[2][1] * y1;
    x[2] = y2 / A[2][2];
    x[1] = (y1 - A[1][2] * x[2]) / A[1][1];
    x[0] = ((y0 - A[0][1] * x[1]) - A[0][2] * x[2]) / A[0][0];
}
SGX_DEV void sgx_sim3_log(const SgxSim3 &S, double res[7])
{   // Sim3::log, sim3.h:145-226
    const double sigma = log(S.s);
    double R[3][3]; sgx_quat_to_R(S.q, R);
    const double d = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1);
    const double dR[3] = { R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1] };
    const double eps = 0.00001;
    double f = 0.5, A, B, C;
    if (fabs(sigma) < eps) {
        C = 1;
        if (d > 1 - eps) { A = 1. / 2.; B = 1. / 6.; }
        else { const double theta = acos(d), theta2 = theta * theta; f = theta / (2 * sqrt(1 - d * d)); A = (1 - cos(theta)) / theta2; B = (theta - sin(theta)) / (theta2 * theta); }
    } else {
        C = (S.s - 1) / sigma;
        if (d > 1 - eps) { const double sigma2 = sigma * sigma; A = ((sigma - 1) * S.s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * S.s) / (sigma2 * sigma); }
        else {
            const double theta = acos(d); f = theta / (2 * sqrt(1 - d * d));
            const double theta2 = theta * theta, a = S.s * sin(theta), b = S.s * cos(theta), c = theta2 + sigma * sigma;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    const double w0 = f * dR[0], w1 = f * dR[1], w2 = f * dR[2];
    const double O[3][3] = { { 0, -w2, w1 }, { w2, 0, -w0 }, { -w1, w0, 0 } };
    double O2[3][3], W[3][3], ups[3];
    sgx_mat3_mul(O, O, O2);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) W[i][j] = (A * O[i][j] + B * O2[i][j]) + C * (i == j);
    sgx_lu3_solve(W, S.t, ups);
    res[0] = w0; res[1] = w1; res[2] = w2; res[3] = ups[0]; res[4] = ups[1]; res[5] = ups[2]; res[6] = sigma;
}
