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
#include "orc_se3.h"

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

/* linearizeOplus of Edge(Stereo)SE3ProjectXYZOnlyPose, types_six_dof_expmap.cpp:266-288, 335-364 (a mono edge has no third row: left untouched) */
static void pedge_jacobian(int stereo, const se3q *T, const double *Xw, const pcam *cam, double J[3][6])
{
    double p[3]; se3_map(T, Xw, p);
    const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
    J[0][0] = x * y * invz_2 * cam->fx; J[0][1] = -(1 + (x * x * invz_2)) * cam->fx; J[0][2] = y * invz * cam->fx;
    J[0][3] = -invz * cam->fx; J[0][4] = 0; J[0][5] = x * invz_2 * cam->fx;
    J[1][0] = (1 + y * y * invz_2) * cam->fy; J[1][1] = -x * y * invz_2 * cam->fy; J[1][2] = -x * invz * cam->fy;
    J[1][3] = 0; J[1][4] = -invz * cam->fy; J[1][5] = y * invz_2 * cam->fy;
    if (stereo) {
        J[2][0] = J[0][0] - cam->bf * y * invz_2; J[2][1] = J[0][1] + cam->bf * x * invz_2; J[2][2] = J[0][2];
        J[2][3] = J[0][3]; J[2][4] = 0; J[2][5] = J[0][5] - cam->bf * invz_2;
    }
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
                double J[3][6];
                pedge_jacobian(e->stereo, &est, e->Xw, &cam, J);
                const int D = e->stereo ? 3 : 2;
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

/* ---- known-answer tap (tests/test_oracle_kat.py): error and analytic Jacobian of one pose-only edge at exp(delta) * Tcw ---- */
int orc_kat_pose_edge(const float *Tcw, const double *Xw, const double *obs, int stereo, double fx, double fy, double cx, double cy, double bf,
                      const double *delta, double *err, double *Jout)
{
    se3q T, ex, Tp; se3_from_cv(Tcw, &T); se3_exp(delta, &ex); se3_mul(&ex, &T, &Tp);
    pcam cam = { fx, fy, cx, cy, bf };
    pedge e; memset(&e, 0, sizeof e); e.stereo = stereo; e.info = 1;
    for (int d = 0; d < 3; d++) { e.obs[d] = obs[d]; e.Xw[d] = Xw[d]; }
    edge_error(&e, &Tp, &cam);
    double J[3][6]; memset(J, 0, sizeof J);
    pedge_jacobian(stereo, &Tp, e.Xw, &cam, J);
    for (int d = 0; d < 3; d++) { err[d] = e.err[d]; for (int c = 0; c < 6; c++) Jout[6 * d + c] = J[d][c]; }
    return 0;
}
