/* oracle/sim3_oracle.c — CPU ORACLE for Optimizer::OptimizeSim3.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates (fp64, fp32 only at the cv::Mat boundary, as the reference):
 *   Optimizer::OptimizeSim3                      src/sg-slam/src/Optimizer.cc:1046-1257   (caller LoopClosing::ComputeSim3, LoopClosing.cc:326)
 * and the vendored g2o pieces it executes (G = src/sg-slam/Thirdparty/g2o/g2o):
 *   Sim3 (exp constructor, *, inverse, map)       G/types/sim3.h:72-139, :230-289
 *   VertexSim3Expmap::oplusImpl, cam_map1/2       G/types/types_seven_dof_expmap.h:58-86
 *   Edge(Inverse)Sim3ProjectXYZ::computeError     G/types/types_seven_dof_expmap.h:131-171 — linearizeOplus is commented out there, so the
 *   NUMERIC Jacobian of BaseBinaryEdge applies    G/core/base_binary_edge.hpp:131-205 (central differences, delta 1e-9, per dimension through oplus)
 *   OptimizationAlgorithmLevenberg::solve, BlockSolver, RobustKernelHuber, LinearSolverDense (Eigen LDLT): as in poseopt_oracle.c
 * Eigen operations (Quaterniond(Matrix3d), q*v, q*q, LDLT) restated from Eigen's published algorithms: PARITY UNPINNED at that boundary. */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc_se3.h"

typedef struct { double q[4]; /* x,y,z,w */ double t[3]; double s; } sim3;

static void sim3_exp(const double u[7], sim3 *o)
{   /* Sim3(const Vector7d &update), sim3.h:72-139 */
    const double w[3] = { u[0], u[1], u[2] }, up[3] = { u[3], u[4], u[5] }, sigma = u[6];
    const double theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double O[3][3] = { { 0, -w[2], w[1] }, { w[2], 0, -w[0] }, { -w[1], w[0], 0 } }, O2[3][3], R[3][3];
    mat3_mul(O, O, O2);
    o->s = exp(sigma);
    const double eps = 0.00001;
    double A, B, C;
    if (fabs(sigma) < eps) {
        C = 1;
        if (theta < eps) {
            A = 1. / 2.; B = 1. / 6.;
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = ((i == j) + O[i][j]) + O2[i][j];
        } else {
            const double theta2 = theta * theta;
            A = (1 - cos(theta)) / theta2; B = (theta - sin(theta)) / (theta2 * theta);
            const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = ((i == j) + a * O[i][j]) + b * O2[i][j];
        }
    } else {
        C = (o->s - 1) / sigma;
        if (theta < eps) {
            const double sigma2 = sigma * sigma;
            A = ((sigma - 1) * o->s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * o->s) / (sigma2 * sigma);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = ((i == j) + O[i][j]) + O2[i][j];
        } else {
            const double sa = sin(theta) / theta, sb = (1 - cos(theta)) / (theta * theta);
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i][j] = ((i == j) + sa * O[i][j]) + sb * O2[i][j];
            const double a = o->s * sin(theta), b = o->s * cos(theta), theta2 = theta * theta, sigma2 = sigma * sigma, c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    quat_from_R(R, o->q);
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += ((A * O[i][k] + B * O2[i][k]) + C * (i == k)) * up[k];
        o->t[i] = s;
    }
}
static void sim3_map(const sim3 *S, const double x[3], double o[3]) { double r[3]; quat_rotate(S->q, x, r); for (int i = 0; i < 3; i++) o[i] = S->s * r[i] + S->t[i]; }
static void sim3_mul(const sim3 *a, const sim3 *b, sim3 *o)
{   /* operator*, sim3.h:262-268 */
    sim3 r; quat_mul(a->q, b->q, r.q);
    double rt[3]; quat_rotate(a->q, b->t, rt);
    for (int i = 0; i < 3; i++) r.t[i] = a->s * rt[i] + a->t[i];
    r.s = a->s * b->s; *o = r;
}
static void sim3_inverse(const sim3 *a, sim3 *o)
{   /* Sim3(r.conjugate(), r.conjugate()*((-1./s)*t), 1./s), sim3.h:230-233 */
    sim3 r; r.q[0] = -a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = a->q[3];
    const double f = -1. / a->s; const double v[3] = { f * a->t[0], f * a->t[1], f * a->t[2] };
    quat_rotate(r.q, v, r.t); r.s = 1. / a->s; *o = r;
}

typedef struct { double p[3], obs[2], info, err[2]; int inverse, alive; } sedge;
typedef struct { double f1[2], c1[2], f2[2], c2[2]; } scam;
static void sedge_error(const sedge *e, const sim3 *S, const sim3 *Sinv, const scam *K, double err[2])
{
    double m[3];
    if (!e->inverse) { sim3_map(S, e->p, m); err[0] = e->obs[0] - (m[0] / m[2] * K->f1[0] + K->c1[0]); err[1] = e->obs[1] - (m[1] / m[2] * K->f1[1] + K->c1[1]); }
    else { sim3_map(Sinv, e->p, m); err[0] = e->obs[0] - (m[0] / m[2] * K->f2[0] + K->c2[0]); err[1] = e->obs[1] - (m[1] / m[2] * K->f2[1] + K->c2[1]); }
}
static double sedge_chi2(const sedge *e) { return e->err[0] * (e->info * e->err[0]) + e->err[1] * (e->info * e->err[1]); }
static void shuber(double e, double delta, double rho[3])
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double sq = sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
}
static void oplus(const sim3 *est, const double upd[7], int fix_scale, sim3 *o)
{   /* VertexSim3Expmap::oplusImpl */
    double u[7]; memcpy(u, upd, sizeof u);
    if (fix_scale) u[6] = 0;
    sim3 d; sim3_exp(u, &d); sim3_mul(&d, est, o);
}

/* Eigen::LDLT-style factorisation with diagonal pivoting of a 7x7 (poseopt_oracle.c ldlt6_solve, n = 7); returns 0 when "not positive" */
static int ldlt7_solve(const double Hin[7][7], const double b[7], double x[7])
{
    enum { N = 7 };
    double A[N][N]; int perm[N];
    memcpy(A, Hin, sizeof A);
    for (int i = 0; i < N; i++) perm[i] = i;
    int sign = 0;
    for (int k = 0; k < N; k++) {
        int p = k; double big = fabs(A[k][k]);
        for (int i = k + 1; i < N; i++) if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
        if (k == 0) sign = A[p][p] > 0 ? 1 : -1;
        if (p != k) {
            for (int j = 0; j < N; j++) { double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
            for (int i = 0; i < N; i++) { double t = A[i][k]; A[i][k] = A[i][p]; A[i][p] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        const double d = A[k][k];
        if (!(d == d) || d == 0) { if (!(d == d)) return 0; continue; }
        for (int i = k + 1; i < N; i++) {
            const double l = A[i][k] / d;
            for (int j = k + 1; j < N; j++) A[i][j] -= l * A[k][j];
            A[i][k] = l;
        }
    }
    if (sign != 1) return 0;
    double y[N];
    for (int i = 0; i < N; i++) y[i] = b[perm[i]];
    for (int i = 0; i < N; i++) for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
    for (int i = 0; i < N; i++) { if (A[i][i] != 0) y[i] /= A[i][i]; else y[i] = 0; }
    for (int i = N - 1; i >= 0; i--) for (int j = i + 1; j < N; j++) y[i] -= A[j][i] * y[j];
    for (int i = 0; i < N; i++) x[perm[i]] = y[i];
    return 1;
}

/* one optimizer.optimize(iterations) call over the alive edges; returns the iterations run */
static int sim3_optimize(sedge *E, int ne, sim3 *est, const scam *K, double delta, int fix_scale, int iterations)
{
    double lambda = -1, ni = 2; int nBadLM = 0, iters = 0;
    for (int it = 0; it < iterations; it++) {
        sim3 inv; sim3_inverse(est, &inv);
        double currentChi = 0;
        for (int k = 0; k < ne; k++) { sedge *e = &E[k]; if (!e->alive) continue; sedge_error(e, est, &inv, K, e->err); double r[3]; shuber(sedge_chi2(e), delta, r); currentChi += r[0]; }
        double tempChi = currentChi; const double iniChi = currentChi;
        /* buildSystem: numeric Jacobians w.r.t. the Sim3 vertex (the point vertices are fixed), base_binary_edge.hpp:147-197 */
        sim3 Sp[7], Sm[7], Spi[7], Smi[7];
        const double dl = 1e-9, scalar = 1.0 / (2 * dl);
        for (int d = 0; d < 7; d++) {
            double add[7] = { 0, 0, 0, 0, 0, 0, 0 };
            add[d] = dl; oplus(est, add, fix_scale, &Sp[d]); sim3_inverse(&Sp[d], &Spi[d]);
            add[d] = -dl; oplus(est, add, fix_scale, &Sm[d]); sim3_inverse(&Sm[d], &Smi[d]);
        }
        double H[7][7], b[7];
        memset(H, 0, sizeof H); memset(b, 0, sizeof b);
        for (int k = 0; k < ne; k++) {
            sedge *e = &E[k]; if (!e->alive) continue;
            double J[2][7];
            for (int d = 0; d < 7; d++) {
                double ep[2], em[2];
                sedge_error(e, &Sp[d], &Spi[d], K, ep); sedge_error(e, &Sm[d], &Smi[d], K, em);
                J[0][d] = scalar * (ep[0] - em[0]); J[1][d] = scalar * (ep[1] - em[1]);
            }
            double r[3]; shuber(sedge_chi2(e), delta, r);
            const double rho1 = r[1], w = rho1 * e->info;
            for (int a = 0; a < 7; a++) {
                const double s = J[0][a] * (e->info * e->err[0]) + J[1][a] * (e->info * e->err[1]);
                b[a] -= rho1 * s;
                for (int c = 0; c < 7; c++) H[a][c] += J[0][a] * w * J[0][c] + J[1][a] * w * J[1][c];
            }
        }
        if (it == 0) { double maxd = 0; for (int j = 0; j < 7; j++) if (fabs(H[j][j]) > maxd) maxd = fabs(H[j][j]); lambda = 1e-5 * maxd; ni = 2; nBadLM = 0; }
        double rho = 0; int qmax = 0;
        do {
            const sim3 backup = *est;
            double Hl[7][7]; memcpy(Hl, H, sizeof H);
            for (int j = 0; j < 7; j++) Hl[j][j] += lambda;
            double x[7] = { 0, 0, 0, 0, 0, 0, 0 };
            const int ok2 = ldlt7_solve(Hl, b, x);
            sim3 upd; oplus(est, x, fix_scale, &upd); *est = upd;
            sim3 inv2; sim3_inverse(est, &inv2);
            tempChi = 0;
            for (int k = 0; k < ne; k++) { sedge *e = &E[k]; if (!e->alive) continue; sedge_error(e, est, &inv2, K, e->err); double r[3]; shuber(sedge_chi2(e), delta, r); tempChi += r[0]; }
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0; for (int j = 0; j < 7; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3; rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                const double sf = alpha > 1. / 3. ? alpha : 1. / 3.;
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else { lambda *= ni; ni *= 2; *est = backup; }            /* pop: the edges keep the rejected trial's errors */
            qmax++;
        } while (rho < 0 && qmax < 10);
        iters = it + 1;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) break;
    }
    return iters;
}

/* Optimizer::OptimizeSim3.  n valid correspondences (the caller lists only pairs whose two map points exist, are not bad and are observed in pKF2, Optimizer.cc:1107-1133, in index
 * order).  p1c / p2c: the two map points in THEIR OWN keyframe's camera frame (R1w*P3D1w + t1w, R2w*P3D2w + t2w: float, then widened); obs1 / obs2: kpUn.pt; info1 / info2: the
 * keypoints' mvInvLevelSigma2; K1 / K2 = (fx, fy, cx, cy).  S12 in/out = (qx, qy, qz, qw, tx, ty, tz, s).  inlier[i] (out) = 0 where vpMatches1[idx] is set to NULL.
 * iters (optional, 2 ints).  Returns nIn (0 on the "fewer than 10 survivors" exit, which leaves S12 untouched). */
int orc_optimize_sim3(int n, const float *p1c, const float *p2c, const float *obs1, const float *obs2, const float *info1, const float *info2,
                      const float *K1, const float *K2, double *S12, float th2, int fix_scale, uint8_t *inlier, int *iters)
{
    sedge *E = (sedge *)malloc(sizeof(sedge) * (size_t)(2 * n > 0 ? 2 * n : 1));
    scam K = { { K1[0], K1[1] }, { K1[2], K1[3] }, { K2[0], K2[1] }, { K2[2], K2[3] } };
    for (int i = 0; i < n; i++) {
        sedge *a = &E[2 * i], *b = &E[2 * i + 1];
        memset(a, 0, sizeof *a); memset(b, 0, sizeof *b);
        for (int c = 0; c < 3; c++) { a->p[c] = p2c[3 * i + c]; b->p[c] = p1c[3 * i + c]; }      /* e12: x1 = S12 * X2;  e21: x2 = S21 * X1 */
        a->obs[0] = obs1[2 * i]; a->obs[1] = obs1[2 * i + 1]; a->info = info1[i]; a->inverse = 0; a->alive = 1;
        b->obs[0] = obs2[2 * i]; b->obs[1] = obs2[2 * i + 1]; b->info = info2[i]; b->inverse = 1; b->alive = 1;
        inlier[i] = 1;
    }
    sim3 est; memcpy(est.q, S12, 32); memcpy(est.t, S12 + 4, 24); est.s = S12[7];
    const float deltaHuber = sqrtf(th2);                        /* const float deltaHuber = sqrt(th2) */
    int it0 = sim3_optimize(E, 2 * n, &est, &K, (double)deltaHuber, fix_scale, 5);
    int nBad = 0;
    for (int i = 0; i < n; i++)
        if (sedge_chi2(&E[2 * i]) > th2 || sedge_chi2(&E[2 * i + 1]) > th2) { inlier[i] = 0; E[2 * i].alive = 0; E[2 * i + 1].alive = 0; nBad++; }
    if (iters) { iters[0] = it0; iters[1] = 0; }
    const int more = nBad > 0 ? 10 : 5;
    if (n - nBad < 10) { free(E); return 0; }
    int it1 = sim3_optimize(E, 2 * n, &est, &K, (double)deltaHuber, fix_scale, more);
    if (iters) iters[1] = it1;
    int nIn = 0;
    for (int i = 0; i < n; i++) {
        if (!E[2 * i].alive) continue;
        if (sedge_chi2(&E[2 * i]) > th2 || sedge_chi2(&E[2 * i + 1]) > th2) inlier[i] = 0; else nIn++;
    }
    memcpy(S12, est.q, 32); memcpy(S12 + 4, est.t, 24); S12[7] = est.s;
    free(E);
    return nIn;
}

/* ===================================================================================================================
 * Optimizer::OptimizeEssentialGraph (src/sg-slam/src/Optimizer.cc:781-1042; caller LoopClosing::CorrectLoop, LoopClosing.cc:567): the pose-graph optimisation itself.
 * The graph construction (:807-958 — which keyframes are connected and the measurements Sji = Sjw * Swi) walks the map's covisibility / spanning-tree / loop-edge containers and
 * stays with the caller; this function takes the vertices (initial Sim3 estimates, the fixed loop keyframe) and the EdgeSim3 list in insertion order and runs
 *   solver->setUserLambdaInit(1e-16); optimizer.initializeOptimization(); optimizer.optimize(20);                                  (:794, :961-962)
 * EdgeSim3::computeError: _error = (C * v1 * v2^-1).log() (G/types/types_seven_dof_expmap.h:100-108), information = identity, no robust kernel, numeric Jacobians for both
 * vertices (G/core/base_binary_edge.hpp:131-205); Sim3::log G/types/sim3.h:145-226 (deltaR: G/types/se3_ops.hpp; W.lu().solve(t): Eigen PartialPivLU, restated).
 * The reference solves the sparse normal equations with Eigen's sparse Cholesky (LinearSolverEigen); here they are dense (same solution up to rounding).
 * =================================================================================================================== */
static void lu3_solve(const double Ain[3][3], const double b[3], double x[3])
{   /* Eigen PartialPivLU of a 3x3 and solve */
    double A[3][3]; int p[3] = { 0, 1, 2 };
    memcpy(A, Ain, sizeof A);
    for (int k = 0; k < 3; k++) {
        int piv = k; double big = fabs(A[k][k]);
        for (int i = k + 1; i < 3; i++) if (fabs(A[i][k]) > big) { big = fabs(A[i][k]); piv = i; }
        if (piv != k) { for (int j = 0; j < 3; j++) { double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; } int t = p[k]; p[k] = p[piv]; p[piv] = t; }
        for (int i = k + 1; i < 3; i++) { A[i][k] /= A[k][k]; for (int j = k + 1; j < 3; j++) A[i][j] -= A[i][k] * A[k][j]; }
    }
    double y[3];
    for (int i = 0; i < 3; i++) { y[i] = b[p[i]]; for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j]; }
    for (int i = 2; i >= 0; i--) { double v = y[i]; for (int j = i + 1; j < 3; j++) v -= A[i][j] * x[j]; x[i] = v / A[i][i]; }
}
static void sim3_log(const sim3 *S, double res[7])
{   /* Sim3::log, sim3.h:145-226 */
    const double sigma = log(S->s);
    double R[3][3]; quat_to_R(S->q, R);
    const double d = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1);
    const double dR[3] = { R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1] };
    const double eps = 0.00001;
    double omega[3], A, B, C;
    if (fabs(sigma) < eps) {
        C = 1;
        if (d > 1 - eps) { for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i]; A = 1. / 2.; B = 1. / 6.; }
        else {
            const double theta = acos(d), theta2 = theta * theta, f = theta / (2 * sqrt(1 - d * d));
            for (int i = 0; i < 3; i++) omega[i] = f * dR[i];
            A = (1 - cos(theta)) / theta2; B = (theta - sin(theta)) / (theta2 * theta);
        }
    } else {
        C = (S->s - 1) / sigma;
        if (d > 1 - eps) {
            const double sigma2 = sigma * sigma;
            for (int i = 0; i < 3; i++) omega[i] = 0.5 * dR[i];
            A = ((sigma - 1) * S->s + 1) / sigma2; B = ((0.5 * sigma2 - sigma + 1) * S->s) / (sigma2 * sigma);
        } else {
            const double theta = acos(d), f = theta / (2 * sqrt(1 - d * d));
            for (int i = 0; i < 3; i++) omega[i] = f * dR[i];
            const double theta2 = theta * theta, a = S->s * sin(theta), b = S->s * cos(theta), c = theta2 + sigma * sigma;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / c) * 1. / theta2;
        }
    }
    const double O[3][3] = { { 0, -omega[2], omega[1] }, { omega[2], 0, -omega[0] }, { -omega[1], omega[0], 0 } };
    double O2[3][3], W[3][3], ups[3];
    mat3_mul(O, O, O2);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W[i][j] = (A * O[i][j] + B * O2[i][j]) + C * (i == j);
    lu3_solve(W, S->t, ups);
    for (int i = 0; i < 3; i++) { res[i] = omega[i]; res[3 + i] = ups[i]; }
    res[6] = sigma;
}
static void eg_error(const sim3 *C, const sim3 *vi, const sim3 *vj, double e[7])
{
    sim3 a, inv, b; sim3_mul(C, vi, &a); sim3_inverse(vj, &inv); sim3_mul(&a, &inv, &b); sim3_log(&b, e);
}
static void load_sim3(const double *p, sim3 *s) { memcpy(s->q, p, 32); memcpy(s->t, p + 4, 24); s->s = p[7]; }
static void store_sim3(const sim3 *s, double *p) { memcpy(p, s->q, 32); memcpy(p + 4, s->t, 24); p[7] = s->s; }

/* dense Cholesky solve of the n x n system (row-major, only the full symmetric matrix is read); returns 0 when a pivot is not positive */
static int chol_solve_dense(int n, double *A, const double *b, double *x)
{
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0)) return 0;
        d = sqrt(d); A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double v = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = v / d;
        }
    }
    for (int i = 0; i < n; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= A[(size_t)i * n + k] * x[k]; x[i] = v / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= A[(size_t)k * n + i] * x[k]; x[i] = v / A[(size_t)i * n + i]; }
    return 1;
}

/* S (nv x 8, in/out): vertex estimates (qx,qy,qz,qw,tx,ty,tz,s); fixed[v] != 0: setFixed(true); edges e: vertex 0 = e_i, vertex 1 = e_j, measurement e_meas (ne x 8).
 * stats (optional, 3 doubles): iterations run, chi2 before, chi2 after.  Returns the number of LM iterations. */
int orc_optimize_essential_graph(int nv, double *S, const uint8_t *fixed, int ne, const int *e_i, const int *e_j, const double *e_meas, int fix_scale, int iterations, double *stats)
{
    sim3 *V = (sim3 *)malloc(sizeof(sim3) * (size_t)(nv > 0 ? nv : 1)), *Bk = (sim3 *)malloc(sizeof(sim3) * (size_t)(nv > 0 ? nv : 1));
    int *hidx = (int *)malloc(sizeof(int) * (size_t)(nv > 0 ? nv : 1));
    int nf = 0;
    for (int v = 0; v < nv; v++) { load_sim3(S + 8 * v, &V[v]); hidx[v] = fixed[v] ? -1 : nf++; }
    const int n = 7 * nf;
    double *H = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (size_t)(n > 0 ? n : 1)), *Hl = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1) * (size_t)(n > 0 ? n : 1));
    double *b = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)), *x = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    double (*err)[7] = (double (*)[7])malloc(sizeof(double) * 7 * (size_t)(ne > 0 ? ne : 1));
    sim3 *M = (sim3 *)malloc(sizeof(sim3) * (size_t)(ne > 0 ? ne : 1));
    for (int k = 0; k < ne; k++) load_sim3(e_meas + 8 * k, &M[k]);
    double lambda = -1, ni = 2; int nBadLM = 0, iters = 0; double chi_first = 0, chi_last = 0;
    for (int it = 0; it < iterations && n > 0; it++) {
        double currentChi = 0;
        for (int k = 0; k < ne; k++) { eg_error(&M[k], &V[e_i[k]], &V[e_j[k]], err[k]); for (int c = 0; c < 7; c++) currentChi += err[k][c] * err[k][c]; }
        if (it == 0) chi_first = currentChi;
        double tempChi = currentChi; const double iniChi = currentChi;
        memset(H, 0, sizeof(double) * (size_t)n * n); memset(b, 0, sizeof(double) * (size_t)n);
        const double dl = 1e-9, scalar = 1.0 / (2 * dl);
        for (int k = 0; k < ne; k++) {
            const int vi = e_i[k], vj = e_j[k], hi = hidx[vi], hj = hidx[vj];
            double J[2][7][7];                                     /* J[side][row][col] */
            for (int side = 0; side < 2; side++) {
                const int hv = side ? hj : hi; if (hv < 0) continue;
                for (int d = 0; d < 7; d++) {
                    double add[7] = { 0, 0, 0, 0, 0, 0, 0 }, ep[7], em[7]; sim3 pp, pm;
                    add[d] = dl; oplus(side ? &V[vj] : &V[vi], add, fix_scale, &pp);
                    add[d] = -dl; oplus(side ? &V[vj] : &V[vi], add, fix_scale, &pm);
                    if (!side) { eg_error(&M[k], &pp, &V[vj], ep); eg_error(&M[k], &pm, &V[vj], em); } else { eg_error(&M[k], &V[vi], &pp, ep); eg_error(&M[k], &V[vi], &pm, em); }
                    for (int r = 0; r < 7; r++) J[side][r][d] = scalar * (ep[r] - em[r]);
                }
            }
            for (int sa = 0; sa < 2; sa++) {
                const int ha = sa ? hj : hi; if (ha < 0) continue;
                for (int a = 0; a < 7; a++) {
                    double s = 0; for (int r = 0; r < 7; r++) s += J[sa][r][a] * err[k][r];
                    b[7 * ha + a] -= s;
                    for (int sb = 0; sb < 2; sb++) {
                        const int hb = sb ? hj : hi; if (hb < 0) continue;
                        for (int c = 0; c < 7; c++) { double h = 0; for (int r = 0; r < 7; r++) h += J[sa][r][a] * J[sb][r][c]; H[(size_t)(7 * ha + a) * n + 7 * hb + c] += h; }
                    }
                }
            }
        }
        if (it == 0) { lambda = 1e-16; ni = 2; nBadLM = 0; }          /* computeLambdaInit: _userLambdaInit > 0 */
        double rho = 0; int qmax = 0;
        do {
            memcpy(Bk, V, sizeof(sim3) * (size_t)nv);
            memcpy(Hl, H, sizeof(double) * (size_t)n * n);
            for (int j = 0; j < n; j++) Hl[(size_t)j * n + j] += lambda;
            const int ok2 = chol_solve_dense(n, Hl, b, x);           /* a failed factorisation leaves x from the previous solve, as g2o's _x */
            for (int v = 0; v < nv; v++) if (hidx[v] >= 0) { sim3 u; oplus(&V[v], x + 7 * hidx[v], fix_scale, &u); V[v] = u; }
            tempChi = 0;
            for (int k = 0; k < ne; k++) { eg_error(&M[k], &V[e_i[k]], &V[e_j[k]], err[k]); for (int c = 0; c < 7; c++) tempChi += err[k][c] * err[k][c]; }
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0; for (int j = 0; j < n; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3; rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                const double sf = alpha > 1. / 3. ? alpha : 1. / 3.;
                lambda *= sf; ni = 2; currentChi = tempChi;
            } else { lambda *= ni; ni *= 2; memcpy(V, Bk, sizeof(sim3) * (size_t)nv); }
            qmax++;
        } while (rho < 0 && qmax < 10);
        iters = it + 1; chi_last = currentChi;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) break;
    }
    for (int v = 0; v < nv; v++) store_sim3(&V[v], S + 8 * v);
    if (stats) { stats[0] = iters; stats[1] = chi_first; stats[2] = chi_last; }
    free(V); free(Bk); free(hidx); free(H); free(Hl); free(b); free(x); free(err); free(M);
    return iters;
}

/* the map-point correction that follows (Optimizer.cc:1004-1041): P' = correctedSwr.map(Srw.map(P)) with the point's reference keyframe r; fp64 inside, float at the cv::Mat boundary */
void orc_correct_map_points(int n, const float *xw, const int *ref, const double *Srw, const double *corrected_Swr, float *out)
{
    for (int i = 0; i < n; i++) {
        sim3 a, c; load_sim3(Srw + 8 * ref[i], &a); load_sim3(corrected_Swr + 8 * ref[i], &c);
        const double p[3] = { xw[3 * i], xw[3 * i + 1], xw[3 * i + 2] }; double m[3], o[3];
        sim3_map(&a, p, m); sim3_map(&c, m, o);
        out[3 * i] = (float)o[0]; out[3 * i + 1] = (float)o[1]; out[3 * i + 2] = (float)o[2];
    }
}

/* known-answer taps (tests/test_sim3.py): Sim3 exp / log / product / inverse on (qx,qy,qz,qw,tx,ty,tz,s) */
void orc_kat_sim3_exp(const double *u, double *out8) { sim3 s; sim3_exp(u, &s); store_sim3(&s, out8); }
void orc_kat_sim3_log(const double *s8, double *out7) { sim3 s; load_sim3(s8, &s); sim3_log(&s, out7); }
