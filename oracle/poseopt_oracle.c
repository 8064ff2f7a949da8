/* oracle/poseopt_oracle.c — CPU ORACLE for Optimizer::PoseOptimization.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates (fp64 throughout, fp32 only at the cv::Mat boundary, as the reference):
 *   Optimizer::PoseOptimization                 src/sg-slam/src/Optimizer.cc:239-451
 *   Converter::toSE3Quat / toCvMat              src/sg-slam/src/Converter.cc:37-71
 * and the vendored g2o pieces it executes (G = src/sg-slam/Thirdparty/g2o/g2o):
 *   OptimizationAlgorithmLevenberg::solve       G/core/optimization_algorithm_levenberg.cpp:61-189
 *   SparseOptimizer::optimize / update          G/core/sparse_optimizer.cpp:354-435
 *   BlockSolver buildSystem/setLambda/solve     G/core/block_solver.hpp:354-366,502-604
 *   BaseUnaryEdge::constructQuadraticForm       G/core/base_unary_edge.hpp:43-72
 *   RobustKernelHuber::robustify                G/core/robust_kernel_impl.cpp:78-91
 *   Edge(Stereo)SE3ProjectXYZOnlyPose           G/types/types_six_dof_expmap.h:146-202, .cpp:266-364
 *   SE3Quat (exp, operator*, map, normalize)    G/types/se3quat.h:41-296,  skew G/types/se3_ops.hpp:27-38
 *   LinearSolverDense (Eigen LDLT, isPositive)  G/solvers/linear_solver_dense.h:65-113
 * Eigen (3.1, external) operations — Quaterniond(Matrix3d), q*v, q*q, toRotationMatrix, LDLT — are
 * restated from Eigen's published algorithms:   ==> PARITY UNPINNED at the Eigen boundary <==
 * (irrelevant at the 1e-5 relative tolerance the north star sets for pose residuals).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint;
typedef struct { double q[4]; /* x,y,z,w */ double t[3]; } se3q;

static void quat_from_R(const double R[3][3], double q[4])
{   /* Eigen Quaternion = Matrix3 */
    double t = R[0][0] + R[1][1] + R[2][2];
    if (t > 0) {
        t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (R[2][1] - R[1][2]) * t; q[1] = (R[0][2] - R[2][0]) * t; q[2] = (R[1][0] - R[0][1]) * t;
    } else {
        int i = 0; if (R[1][1] > R[0][0]) i = 1; if (R[2][2] > R[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[k][j] - R[j][k]) * t; q[j] = (R[j][i] + R[i][j]) * t; q[k] = (R[k][i] + R[i][k]) * t;
    }
}
static void quat_normalize_rot(double q[4])
{   /* SE3Quat::normalizeRotation se3quat.h:280-285 */
    if (q[3] < 0) for (int i = 0; i < 4; i++) q[i] *= -1;
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}
static void quat_rotate(const double q[4], const double v[3], double o[3])
{   /* Eigen QuaternionBase::_transformVector */
    double uv[3] = { q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    o[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    o[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}
static void quat_mul(const double a[4], const double b[4], double o[4])
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
static void quat_to_R(const double q[4], double R[3][3])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
    R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}
static void se3_map(const se3q *T, const double x[3], double o[3]) { quat_rotate(T->q, x, o); o[0] += T->t[0]; o[1] += T->t[1]; o[2] += T->t[2]; }

static void mat3_mul(const double A[3][3], const double B[3][3], double C[3][3])
{ for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[i][k] * B[k][j]; C[i][j] = s; } }

static void se3_exp(const double u[6], se3q *out)
{   /* SE3Quat::exp se3quat.h:223-257 (omega = u[0..2], upsilon = u[3..5]) */
    const double w[3] = { u[0], u[1], u[2] }, up[3] = { u[3], u[4], u[5] };
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double O[3][3] = { { 0, -w[2], w[1] }, { w[2], 0, -w[0] }, { -w[1], w[0], 0 } };
    double O2[3][3]; mat3_mul(O, O, O2);
    double R[3][3], V[3][3];
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = (i == j) + O[i][j] + O2[i][j]; V[i][j] = R[i][j]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            R[i][j] = (i == j) + a * O[i][j] + b * O2[i][j];
            V[i][j] = (i == j) + b * O[i][j] + c * O2[i][j];
        }
    }
    quat_from_R(R, out->q);
    for (int i = 0; i < 3; i++) out->t[i] = V[i][0] * up[0] + V[i][1] * up[1] + V[i][2] * up[2];
    quat_normalize_rot(out->q);
}
static void se3_mul(const se3q *a, const se3q *b, se3q *o)
{   /* SE3Quat::operator* se3quat.h:104-110 */
    double rt[3]; quat_rotate(a->q, b->t, rt);
    se3q r; r.t[0] = a->t[0] + rt[0]; r.t[1] = a->t[1] + rt[1]; r.t[2] = a->t[2] + rt[2];
    quat_mul(a->q, b->q, r.q); quat_normalize_rot(r.q);
    *o = r;
}
static void se3_from_cv(const float *T, se3q *o)
{   /* Converter::toSE3Quat */
    double R[3][3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[i][j] = T[4 * i + j]; o->t[i] = T[4 * i + 3]; }
    quat_from_R(R, o->q); quat_normalize_rot(o->q);
}
static void se3_to_cv(const se3q *s, float *T)
{   /* Converter::toCvMat(SE3Quat) */
    double R[3][3]; quat_to_R(s->q, R);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = (float)R[i][j]; T[4 * i + 3] = (float)s->t[i]; }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
}

typedef struct { int idx, stereo, level, robust; double obs[3], info, Xw[3], err[3]; } pedge;
typedef struct { double fx, fy, cx, cy, bf; } pcam;

static void edge_error(pedge *e, const se3q *T, const pcam *c)
{
    double p[3]; se3_map(T, e->Xw, p);
    if (!e->stereo) {       /* types_six_dof_expmap.cpp:290-296: project2d then *f + c */
        const double px = p[0] / p[2], py = p[1] / p[2];
        e->err[0] = e->obs[0] - (px * c->fx + c->cx); e->err[1] = e->obs[1] - (py * c->fy + c->cy); e->err[2] = 0;
    } else {                /* :299-306: invz is a FLOAT */
        const float invz = 1.0f / p[2];
        const double r0 = p[0] * invz * c->fx + c->cx, r1 = p[1] * invz * c->fy + c->cy, r2 = r0 - c->bf * invz;
        e->err[0] = e->obs[0] - r0; e->err[1] = e->obs[1] - r1; e->err[2] = e->obs[2] - r2;
    }
}
static double edge_chi2(const pedge *e)
{   /* _error.dot(information()*_error), information = invSigma2 * I */
    const int D = e->stereo ? 3 : 2;
    double s = 0; for (int i = 0; i < D; i++) s += e->err[i] * (e->info * e->err[i]);
    return s;
}
static void huber(double e, double delta, double rho[3])
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double sq = sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
}

/* Eigen::LDLT-style factorisation with diagonal pivoting of a 6x6; returns 0 when "not positive" */
static int ldlt6_solve(const double Hin[6][6], const double b[6], double x[6])
{
    double A[6][6]; int perm[6];
    memcpy(A, Hin, sizeof A);
    for (int i = 0; i < 6; i++) perm[i] = i;
    int sign = 0;
    for (int k = 0; k < 6; k++) {
        int p = k; double big = fabs(A[k][k]);
        for (int i = k + 1; i < 6; i++) if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
        if (k == 0) sign = A[p][p] > 0 ? 1 : -1;
        if (p != k) {
            for (int j = 0; j < 6; j++) { double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
            for (int i = 0; i < 6; i++) { double t = A[i][k]; A[i][k] = A[i][p]; A[i][p] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        const double d = A[k][k];
        if (!(d == d) || d == 0) { if (!(d == d)) return 0; continue; }
        for (int i = k + 1; i < 6; i++) {
            const double l = A[i][k] / d;
            for (int j = k + 1; j < 6; j++) A[i][j] -= l * A[k][j];
            A[i][k] = l;
        }
    }
    if (sign != 1) return 0;
    double y[6];
    for (int i = 0; i < 6; i++) y[i] = b[perm[i]];
    for (int i = 0; i < 6; i++) for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
    for (int i = 0; i < 6; i++) { if (A[i][i] != 0) y[i] /= A[i][i]; else y[i] = 0; }
    for (int i = 5; i >= 0; i--) for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
    for (int i = 0; i < 6; i++) x[perm[i]] = y[i];
    return 1;
}

/* Optimizer::PoseOptimization.  keys = mvKeysUn, uright = mvuRight, inv_sigma2 = mvInvLevelSigma2,
 * has_mp/xw = mvpMapPoints (non-NULL / GetWorldPos).  Tcw in/out (4x4 float row-major), outlier out.
 * trace (optional, >= 4*11*3 doubles): per round r, per LM iteration it: {chi2 after iteration, lambda, trials};
 * trace_n[r] = iterations executed in round r.  Returns nInitialCorrespondences - nBad. */
int orc_pose_optimization(int N, const orc_keypoint *keys, const float *uright, const float *inv_sigma2,
                          const uint8_t *has_mp, const float *xw, float fx, float fy, float cx, float cy, float bf,
                          float *Tcw, uint8_t *outlier, double *trace, int *trace_n)
{
    pcam cam = { fx, fy, cx, cy, bf };
    pedge *E = (pedge *)malloc(sizeof(pedge) * (N > 0 ? N : 1));
    int ne = 0;
    const float deltaMono = (float)sqrt(5.991), deltaStereo = (float)sqrt(7.815);     /* Optimizer.cc:272-273 */
    for (int i = 0; i < N; i++) {
        if (!has_mp[i]) continue;
        pedge *e = &E[ne++];
        memset(e, 0, sizeof *e);
        e->idx = i; e->stereo = !(uright[i] < 0); e->level = 0; e->robust = 1;
        e->obs[0] = keys[i].x; e->obs[1] = keys[i].y; e->obs[2] = e->stereo ? uright[i] : 0;
        e->info = inv_sigma2[keys[i].octave];
        e->Xw[0] = xw[3 * i]; e->Xw[1] = xw[3 * i + 1]; e->Xw[2] = xw[3 * i + 2];
        outlier[i] = 0;
    }
    const int nInitial = ne;
    if (nInitial < 3) { free(E); return 0; }
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    se3q est; se3_from_cv(Tcw, &est);
    int nBad = 0;
    for (int round = 0; round < 4; round++) {
        se3_from_cv(Tcw, &est);                             /* :377 — every round restarts from the frame's pose */
        /* ---- optimizer.optimize(10): LM over the level-0 edges */
        double lambda = -1, ni = 2; int nBadLM = 0;
        int iters = 0;
        for (int it = 0; it < 10; it++) {
            /* computeActiveErrors + activeRobustChi2 */
            double currentChi = 0;
            for (int k = 0; k < ne; k++) {
                pedge *e = &E[k]; if (e->level != 0) continue;
                edge_error(e, &est, &cam);
                double c2 = edge_chi2(e);
                if (e->robust) { double rho[3]; huber(c2, e->stereo ? deltaStereo : deltaMono, rho); currentChi += rho[0]; } else currentChi += c2;
            }
            double tempChi = currentChi; const double iniChi = currentChi;
            /* buildSystem */
            double H[6][6], b[6];
            memset(H, 0, sizeof H); memset(b, 0, sizeof b);
            for (int k = 0; k < ne; k++) {
                pedge *e = &E[k]; if (e->level != 0) continue;
                double p[3]; se3_map(&est, e->Xw, p);
                const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
                double J[3][6];
                J[0][0] = x * y * invz_2 * cam.fx; J[0][1] = -(1 + (x * x * invz_2)) * cam.fx; J[0][2] = y * invz * cam.fx;
                J[0][3] = -invz * cam.fx; J[0][4] = 0; J[0][5] = x * invz_2 * cam.fx;
                J[1][0] = (1 + y * y * invz_2) * cam.fy; J[1][1] = -x * y * invz_2 * cam.fy; J[1][2] = -x * invz * cam.fy;
                J[1][3] = 0; J[1][4] = -invz * cam.fy; J[1][5] = y * invz_2 * cam.fy;
                const int D = e->stereo ? 3 : 2;
                if (e->stereo) {
                    J[2][0] = J[0][0] - cam.bf * y * invz_2; J[2][1] = J[0][1] + cam.bf * x * invz_2; J[2][2] = J[0][2];
                    J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - cam.bf * invz_2;
                }
                double rho1 = 1.0;
                if (e->robust) { double rho[3]; huber(edge_chi2(e), e->stereo ? deltaStereo : deltaMono, rho); rho1 = rho[1]; }
                const double w = rho1 * e->info;               /* weightedOmega = rho[1]*information */
                for (int a = 0; a < 6; a++) {
                    double s = 0; for (int d = 0; d < D; d++) s += J[d][a] * (e->info * e->err[d]);
                    b[a] -= rho1 * s;
                    for (int c = 0; c < 6; c++) { double h = 0; for (int d = 0; d < D; d++) h += J[d][a] * w * J[d][c]; H[a][c] += h; }
                }
            }
            if (it == 0) {
                double maxd = 0; for (int j = 0; j < 6; j++) if (fabs(H[j][j]) > maxd) maxd = fabs(H[j][j]);
                lambda = 1e-5 * maxd; ni = 2; nBadLM = 0;
            }
            double rho = 0; int qmax = 0;
            do {
                se3q backup = est;                               /* push */
                double Hl[6][6]; memcpy(Hl, H, sizeof H);
                for (int j = 0; j < 6; j++) Hl[j][j] += lambda;
                double x[6] = { 0, 0, 0, 0, 0, 0 };
                const int ok2 = ldlt6_solve(Hl, b, x);
                se3q ex; se3_exp(x, &ex); se3q upd; se3_mul(&ex, &est, &upd); est = upd;     /* oplus */
                tempChi = 0;
                for (int k = 0; k < ne; k++) {
                    pedge *e = &E[k]; if (e->level != 0) continue;
                    edge_error(e, &est, &cam);
                    double c2 = edge_chi2(e);
                    if (e->robust) { double r3[3]; huber(c2, e->stereo ? deltaStereo : deltaMono, r3); tempChi += r3[0]; } else tempChi += c2;
                }
                if (!ok2) tempChi = DBL_MAX;
                rho = currentChi - tempChi;
                double scale = 0; for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                    const double sf = alpha > 1. / 3. ? alpha : 1. / 3.;
                    lambda *= sf; ni = 2; currentChi = tempChi;   /* discardTop */
                } else {
                    lambda *= ni; ni *= 2; est = backup;          /* pop — edge errors keep the rejected trial's values */
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            iters = it + 1;
            if (trace) { double *t = trace + (round * 11 + it) * 3; t[0] = currentChi; t[1] = lambda; t[2] = qmax; }
            if (qmax == 10 || rho == 0) break;                    /* Terminate */
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        if (trace_n) trace_n[round] = iters;
        /* ---- classify (:383-438) */
        nBad = 0;
        for (int k = 0; k < ne; k++) {
            pedge *e = &E[k];
            if (outlier[e->idx]) edge_error(e, &est, &cam);
            const float chi2 = (float)edge_chi2(e);
            if (chi2 > (e->stereo ? chi2Stereo : chi2Mono)) { outlier[e->idx] = 1; e->level = 1; nBad++; }
            else { outlier[e->idx] = 0; e->level = 0; }
            if (round == 2) e->robust = 0;
        }
        if (ne < 10) break;                                       /* optimizer.edges().size()<10 */
    }
    se3_to_cv(&est, Tcw);
    free(E);
    return nInitial - nBad;
}
