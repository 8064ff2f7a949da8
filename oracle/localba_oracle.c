/* oracle/localba_oracle.c — CPU ORACLE for Optimizer::LocalBundleAdjustment.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates (fp64, fp32 at the cv::Mat boundary):
 *   Optimizer::LocalBundleAdjustment             src/sg-slam/src/Optimizer.cc:453-778 (graph already flattened by the caller)
 * and the vendored g2o it drives (G = src/sg-slam/Thirdparty/g2o/g2o):
 *   OptimizationAlgorithmLevenberg::solve        G/core/optimization_algorithm_levenberg.cpp:61-189
 *   BlockSolver<6,3> buildSystem/setLambda/solve G/core/block_solver.hpp:354-486,502-604 (Schur complement)
 *   BaseBinaryEdge::constructQuadraticForm       G/core/base_binary_edge.hpp:55-120
 *   Edge(Stereo)SE3ProjectXYZ                    G/types/types_six_dof_expmap.h:80-140, .cpp:103-157,188-234
 *   VertexSBAPointXYZ::oplusImpl                 G/types/types_sba.h:52-56
 *   SparseOptimizer::buildIndexMapping           G/core/sparse_optimizer.cpp:166-190 (free poses first, then points)
 * The reduced camera system is solved by the reference with Eigen SimplicialLDLT (G/solvers/linear_solver_eigen.h:94-124: a SPARSE LDL^T);
 * here by an LDL^T in the natural (keyframe) order that stores and touches only the ENVELOPE of the matrix (row i from its first structurally
 * non-zero column to the diagonal: fill-in of an LDL^T never leaves the envelope).  It performs exactly the operations of the dense row-oriented
 * LDL^T that this file used until round 3 minus those on structural zeros (which contribute exact zeros), so its results are bit-identical to the
 * dense solver's (orc_ba_debug_set_dense(1) keeps the dense one for that cross-check, tests/test_localba.py) — and 2 000 keyframes / 50 000
 * landmarks (BASELINE config 4) take seconds instead of hours, which is what lets tools/make_golden.py pin that size (VERDICT r3 'next round' #5).
 * Same solution as Eigen's up to rounding / elimination order.   ==> PARITY UNPINNED at the Eigen boundary <==
 *
 * Flattened problem: poses[np] (Tcw 4x4 float, fixed flag; KeyFrame::mnId order), points[nl] (float3; MapPoint order),
 * edges[ne] in insertion order (outer loop over points, Optimizer.cc:572-653): pose index, point index,
 * obs (u, v, uR; uR < 0 => monocular edge), invSigma2.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "orc_se3.h"

typedef struct { double fx, fy, cx, cy, bf; } bcam;
typedef struct { int pose, point, stereo, level, robust; double obs[3], info, err[3]; } bedge;

static void bhuber(double e, double delta, double rho[3])
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double sq = sqrt(e); rho[0] = 2 * sq * delta - dsqr; rho[1] = delta / sq; rho[2] = -0.5 * rho[1] / e; }
}
static void bedge_error(bedge *e, const se3q *T, const double *X, const bcam *c)
{
    double p[3]; se3_map(T, X, p);
    if (!e->stereo) { e->err[0] = e->obs[0] - (p[0] / p[2] * c->fx + c->cx); e->err[1] = e->obs[1] - (p[1] / p[2] * c->fy + c->cy); e->err[2] = 0; }
    else {
        const float invz = 1.0f / p[2];                    /* types_six_dof_expmap.cpp:150-157 — float */
        const double r0 = p[0] * invz * c->fx + c->cx, r1 = p[1] * invz * c->fy + c->cy, r2 = r0 - c->bf * invz;
        e->err[0] = e->obs[0] - r0; e->err[1] = e->obs[1] - r1; e->err[2] = e->obs[2] - r2;
    }
}
static double bedge_chi2(const bedge *e) { const int D = e->stereo ? 3 : 2; double s = 0; for (int i = 0; i < D; i++) s += e->err[i] * (e->info * e->err[i]); return s; }

/* linearizeOplus of Edge(Stereo)SE3ProjectXYZ: A = d err / d point (3x3), Bj = d err / d pose increment (3x6); rows of a monocular edge's third component are 0 */
static void bedge_jacobians(int stereo, const se3q *T, const double *X, const bcam *cam, double A[3][3], double Bj[3][6])
{
    double p[3]; se3_map(T, X, p);
    double R[3][3]; quat_to_R(T->q, R);
    const double x = p[0], y = p[1], z = p[2], z_2 = z * z, fx = cam->fx, fy = cam->fy, bf = cam->bf;
    if (stereo) {                                                    /* .cpp:188-234 */
        for (int c = 0; c < 3; c++) {
            A[0][c] = -fx * R[0][c] / z + fx * x * R[2][c] / z_2;
            A[1][c] = -fy * R[1][c] / z + fy * y * R[2][c] / z_2;
            A[2][c] = A[0][c] - bf * R[2][c] / z_2;
        }
    } else {                                                         /* .cpp:103-139: -1/z * tmp * R */
        const double tmp[2][3] = { { fx, 0, -x / z * fx }, { 0, fy, -y / z * fy } };
        for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) { double s = 0; for (int q = 0; q < 3; q++) s += tmp[r][q] * R[q][c]; A[r][c] = -1. / z * s; }
        for (int c = 0; c < 3; c++) A[2][c] = 0;
    }
    Bj[0][0] = x * y / z_2 * fx; Bj[0][1] = -(1 + (x * x / z_2)) * fx; Bj[0][2] = y / z * fx; Bj[0][3] = -1. / z * fx; Bj[0][4] = 0; Bj[0][5] = x / z_2 * fx;
    Bj[1][0] = (1 + y * y / z_2) * fy; Bj[1][1] = -x * y / z_2 * fy; Bj[1][2] = -x / z * fy; Bj[1][3] = 0; Bj[1][4] = -1. / z * fy; Bj[1][5] = y / z_2 * fy;
    if (stereo) { Bj[2][0] = Bj[0][0] - bf * y / z_2; Bj[2][1] = Bj[0][1] + bf * x / z_2; Bj[2][2] = Bj[0][2]; Bj[2][3] = Bj[0][3]; Bj[2][4] = 0; Bj[2][5] = Bj[0][5] - bf / z_2; }
    else for (int c = 0; c < 6; c++) Bj[2][c] = 0;
}

/* dense LDL^T solve (n x n, row-major, symmetric); returns 0 when a pivot is not positive */
static int dense_ldlt_solve(double *A, int n, const double *b, double *x)
{
    double *d = (double *)malloc(sizeof(double) * n);
    for (int j = 0; j < n; j++) {
        double v = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) v -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * d[k];
        d[j] = v;
        if (!(v > 0)) { free(d); return 0; }
        for (int i = j + 1; i < n; i++) {
            double w = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) w -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * d[k];
            A[(size_t)i * n + j] = w / v;
        }
    }
    for (int i = 0; i < n; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= A[(size_t)i * n + k] * x[k]; x[i] = v; }
    for (int i = 0; i < n; i++) x[i] /= d[i];
    for (int i = n - 1; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < n; k++) v -= A[(size_t)k * n + i] * x[k]; x[i] = v; }
    free(d);
    return 1;
}

/* envelope (skyline) LDL^T: row i holds columns first[i] .. i at env[ptr[i] .. ptr[i + 1]); same operation order as dense_ldlt_solve */
typedef struct { int n; int *first; size_t *ptr; double *v; } envmat;
static double *env_at(const envmat *M, int i, int k) { return M->v + M->ptr[i] + (size_t)(k - M->first[i]); }
static int env_ldlt_solve(envmat *M, const double *b, double *x)
{
    const int n = M->n;
    double *d = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
    /* rows that reach back to column j, for every j: row i > j with first[i] <= j.  A profile is not monotone (the wrap-around keyframes of a loop reach back to
       column 0), so the candidates are kept as a list of "long" rows plus the band below the diagonal */
    int bw = 0;                                        /* max width among the rows that are not "long" */
    int *lng = (int *)malloc(sizeof(int) * (n > 0 ? n : 1)); int nlng = 0;
    {
        /* a row is "long" if it is wider than 4 x the median width: it then goes through the explicit list */
        int *w = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
        for (int i = 0; i < n; i++) w[i] = i - M->first[i];
        int *srt = (int *)malloc(sizeof(int) * (n > 0 ? n : 1)); memcpy(srt, w, sizeof(int) * n);
        for (int i = 1; i < n; i++) { const int t = srt[i]; int j = i - 1; while (j >= 0 && srt[j] > t) { srt[j + 1] = srt[j]; j--; } srt[j + 1] = t; }      /* insertion sort: widths are nearly sorted */
        const int med = n ? srt[n / 2] : 0, cut = 4 * med + 64;
        for (int i = 0; i < n; i++) { if (w[i] > cut) lng[nlng++] = i; else if (w[i] > bw) bw = w[i]; }
        free(w); free(srt);
    }
    int ok = 1;
    for (int j = 0; j < n && ok; j++) {
        const double *rj = env_at(M, j, 0);            /* rj[k] = entry (j, k), valid for k >= first[j] */
        double v = rj[j];
        for (int k = M->first[j]; k < j; k++) v -= rj[k] * rj[k] * d[k];
        d[j] = v;
        if (!(v > 0)) { ok = 0; break; }
        const int iend = j + bw < n - 1 ? j + bw : n - 1;
        for (int pass = 0; pass < 2; pass++) {
            const int cnt = pass == 0 ? iend - j : nlng;
            for (int q = 0; q < cnt; q++) {
                int i;
                if (pass == 0) i = j + 1 + q;                       /* the band below the diagonal (long rows inside it included) */
                else { i = lng[q]; if (i <= iend) continue; }       /* long rows past it */
                if (M->first[i] > j) continue;
                double *ri = env_at(M, i, 0);
                double wv = ri[j];
                const int k0 = M->first[i] > M->first[j] ? M->first[i] : M->first[j];
                for (int k = k0; k < j; k++) wv -= ri[k] * rj[k] * d[k];
                ri[j] = wv / v;
            }
        }
    }
    if (ok) {
        for (int i = 0; i < n; i++) { const double *ri = env_at(M, i, 0); double v = b[i]; for (int k = M->first[i]; k < i; k++) v -= ri[k] * x[k]; x[i] = v; }
        for (int i = 0; i < n; i++) x[i] /= d[i];
        /* back substitution x_i -= sum_{k > i} L[k][i] x_k in the dense solver's order (k ascending for every i, i descending) */
        for (int i = n - 1; i >= 0; i--) {
            double v = x[i];
            const int kend = i + bw < n - 1 ? i + bw : n - 1;
            for (int k = i + 1; k <= kend; k++) if (M->first[k] <= i) v -= *env_at(M, k, i) * x[k];               /* the band (long rows inside it included) */
            for (int q = 0; q < nlng; q++) { const int k = lng[q]; if (k > kend && M->first[k] <= i) v -= *env_at(M, k, i) * x[k]; }      /* long rows past it, ascending */
            x[i] = v;
        }
    }
    free(d); free(lng);
    return ok;
}
static int g_ba_dense = 0;
void orc_ba_debug_set_dense(int on) { g_ba_dense = on ? 1 : 0; }

typedef struct {
    int np, nl, ne, nfree;
    se3q *T; double *X; int *hidx;          /* hidx[pose] = index among free poses or -1 */
    bedge *E; bcam cam;
    float dMono, dStereo;
    const volatile int *stop;
} ba_t;

/* envelope of the reduced system for the current active edge set: block row i1 starts at the smallest free pose that shares an active landmark with it */
static void env_build(const ba_t *B, envmat *Sp)
{
    const int nf = B->nfree, nl = B->nl, NP = 6 * nf;
    envmat Senv; memset(&Senv, 0, sizeof Senv);
    int *minp = (int *)malloc(sizeof(int) * (nf > 0 ? nf : 1)); for (int i = 0; i < nf; i++) minp[i] = i;
    int *lmin = (int *)malloc(sizeof(int) * (nl > 0 ? nl : 1)); for (int l = 0; l < nl; l++) lmin[l] = nf;
    for (int k = 0; k < B->ne; k++) { const int hp = B->hidx[B->E[k].pose]; if (B->E[k].level == 0 && hp >= 0 && hp < lmin[B->E[k].point]) lmin[B->E[k].point] = hp; }
    for (int k = 0; k < B->ne; k++) { const int hp = B->hidx[B->E[k].pose]; if (B->E[k].level == 0 && hp >= 0 && lmin[B->E[k].point] < minp[hp]) minp[hp] = lmin[B->E[k].point]; }
    Senv.n = NP; Senv.first = (int *)malloc(sizeof(int) * (NP > 0 ? NP : 1)); Senv.ptr = (size_t *)malloc(sizeof(size_t) * (NP + 1));
    Senv.ptr[0] = 0;
    for (int r = 0; r < NP; r++) { Senv.first[r] = 6 * minp[r / 6]; Senv.ptr[r + 1] = Senv.ptr[r] + (size_t)(r - Senv.first[r] + 1); }
    Senv.v = (double *)malloc(sizeof(double) * (Senv.ptr[NP] > 0 ? Senv.ptr[NP] : 1));
    free(minp); free(lmin);
    *Sp = Senv;
}
static void env_free(envmat *M) { free(M->first); free(M->ptr); free(M->v); }

static double ba_active_chi2(ba_t *B, int recompute)
{
    double chi = 0;
    for (int k = 0; k < B->ne; k++) {
        bedge *e = &B->E[k]; if (e->level != 0) continue;
        if (recompute) bedge_error(e, &B->T[e->pose], &B->X[3 * e->point], &B->cam);
        const double c2 = bedge_chi2(e);
        if (e->robust) { double r[3]; bhuber(c2, e->stereo ? B->dStereo : B->dMono, r); chi += r[0]; } else chi += c2;
    }
    return chi;
}

typedef struct { double *Hpp, *bp, *Hll, *bl, *Hpl, *S, *xp, *xl, *Dinv, *coef; uint8_t *pt_active; envmat Senv; } ba_buf;

/* buildSystem (block_solver.hpp:502-560) on the level-0 edges, at the current estimate and the errors last computed */
static void ba_build(ba_t *B, ba_buf *W)
{
    const int nf = B->nfree, nl = B->nl, NP = 6 * nf;
    double *Hpp = W->Hpp, *bp = W->bp, *Hll = W->Hll, *bl = W->bl, *Hpl = W->Hpl, *S = W->S, *xp = W->xp, *xl = W->xl, *Dinv = W->Dinv, *coef = W->coef;
    const uint8_t *pt_active = W->pt_active;
    (void)nf; (void)nl; (void)NP; (void)Hpp; (void)bp; (void)Hll; (void)bl; (void)Hpl; (void)S; (void)xp; (void)xl; (void)Dinv; (void)coef; (void)pt_active;
    /* ---- buildSystem */
    memset(Hpp, 0, sizeof(double) * (size_t)nf * 36); memset(bp, 0, sizeof(double) * NP);
    memset(Hll, 0, sizeof(double) * (size_t)nl * 9); memset(bl, 0, sizeof(double) * (size_t)nl * 3);
    for (int k = 0; k < B->ne; k++) {
        bedge *e = &B->E[k]; double *hpl = Hpl + (size_t)k * 18; memset(hpl, 0, sizeof(double) * 18);
        if (e->level != 0) continue;
        const se3q *T = &B->T[e->pose]; const double *X = &B->X[3 * e->point];
        double A[3][3], Bj[3][6];
        bedge_jacobians(e->stereo, T, X, &B->cam, A, Bj);
        const int D = e->stereo ? 3 : 2;
        double rho1 = 1.0;
        if (e->robust) { double r[3]; bhuber(bedge_chi2(e), e->stereo ? B->dStereo : B->dMono, r); rho1 = r[1]; }
        const double w = rho1 * e->info;
        double om_r[3]; for (int d = 0; d < 3; d++) om_r[d] = -(e->info * e->err[d]) * rho1;
        double *hl = Hll + (size_t)e->point * 9, *bL = bl + (size_t)e->point * 3;
        for (int a = 0; a < 3; a++) {
            double s = 0; for (int d = 0; d < D; d++) s += A[d][a] * om_r[d]; bL[a] += s;
            for (int c = 0; c < 3; c++) { double h = 0; for (int d = 0; d < D; d++) h += A[d][a] * w * A[d][c]; hl[3 * a + c] += h; }
        }
        const int hp = B->hidx[e->pose];
        if (hp >= 0) {
            double *hP = Hpp + (size_t)hp * 36, *bP = bp + 6 * hp;
            for (int a = 0; a < 6; a++) {
                double s = 0; for (int d = 0; d < D; d++) s += Bj[d][a] * om_r[d]; bP[a] += s;
                for (int c = 0; c < 6; c++) { double h = 0; for (int d = 0; d < D; d++) h += Bj[d][a] * w * Bj[d][c]; hP[6 * a + c] += h; }
                for (int c = 0; c < 3; c++) { double h = 0; for (int d = 0; d < D; d++) h += Bj[d][a] * w * A[d][c]; hpl[3 * a + c] = h; }
            }
        }
    }
}

/* one linear solve of the damped normal equations through the Schur complement; returns 0 when the reduced system is not positive definite */
static int ba_solve(ba_t *B, ba_buf *W, double lambda)
{
    const int nf = B->nfree, nl = B->nl, NP = 6 * nf;
    double *Hpp = W->Hpp, *bp = W->bp, *Hll = W->Hll, *bl = W->bl, *Hpl = W->Hpl, *S = W->S, *xp = W->xp, *xl = W->xl, *Dinv = W->Dinv, *coef = W->coef;
    const uint8_t *pt_active = W->pt_active;
    (void)nf; (void)nl; (void)NP; (void)Hpp; (void)bp; (void)Hll; (void)bl; (void)Hpl; (void)S; (void)xp; (void)xl; (void)Dinv; (void)coef; (void)pt_active;
    /* ---- solve with Schur complement (block_solver.hpp:367-486); lambda on both diagonals (:573-587) */
    int ok2 = 1;
    memset(coef, 0, sizeof(double) * NP);
    /* S lives in the envelope of its lower triangle (W->Senv; structure from the active edges, W->env_ok = 0 when the edge set changed); the dense copy only for the cross-check */
    envmat *M = &W->Senv;
    if (g_ba_dense) {
        memset(S, 0, sizeof(double) * (size_t)NP * NP);
        for (int i = 0; i < nf; i++) for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
            S[(size_t)(6 * i + a) * NP + 6 * i + c] = Hpp[(size_t)i * 36 + 6 * a + c] + (a == c ? lambda : 0);
    } else {
        memset(M->v, 0, sizeof(double) * M->ptr[NP]);
        for (int i = 0; i < nf; i++) for (int a = 0; a < 6; a++) for (int c = 0; c <= a; c++)
            *env_at(M, 6 * i + a, 6 * i + c) = Hpp[(size_t)i * 36 + 6 * a + c] + (a == c ? lambda : 0);
    }
    for (int l = 0; l < nl; l++) {
        double *Di = Dinv + (size_t)l * 9;
        if (!pt_active[l]) { memset(Di, 0, sizeof(double) * 9); continue; }
        double M[9]; memcpy(M, Hll + (size_t)l * 9, sizeof M); M[0] += lambda; M[4] += lambda; M[8] += lambda;
        const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
        const double det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0 / det;          /* Eigen Matrix3d::inverse(): cofactors / det */
        Di[0] = c00 * id; Di[1] = (M[2] * M[7] - M[1] * M[8]) * id; Di[2] = (M[1] * M[5] - M[2] * M[4]) * id;
        Di[3] = c01 * id; Di[4] = (M[0] * M[8] - M[2] * M[6]) * id; Di[5] = (M[2] * M[3] - M[0] * M[5]) * id;
        Di[6] = c02 * id; Di[7] = (M[1] * M[6] - M[0] * M[7]) * id; Di[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    }
    /* edges of one point are contiguous or not — handle generally with per-point edge lists */
    {
        int *head = (int *)malloc(sizeof(int) * (nl + 1)); int *lst = (int *)malloc(sizeof(int) * (B->ne > 0 ? B->ne : 1));
        memset(head, 0, sizeof(int) * (nl + 1));
        for (int k = 0; k < B->ne; k++) if (B->E[k].level == 0 && B->hidx[B->E[k].pose] >= 0) head[B->E[k].point + 1]++;
        for (int l = 0; l < nl; l++) head[l + 1] += head[l];
        int *fill = (int *)calloc(nl > 0 ? nl : 1, sizeof(int));
        for (int k = 0; k < B->ne; k++) if (B->E[k].level == 0 && B->hidx[B->E[k].pose] >= 0) { const int l = B->E[k].point; lst[head[l] + fill[l]++] = k; }
        for (int l = 0; l < nl; l++) {
            const double *Di = Dinv + (size_t)l * 9;
            double db[3]; for (int a = 0; a < 3; a++) db[a] = Di[3 * a] * bl[3 * l] + Di[3 * a + 1] * bl[3 * l + 1] + Di[3 * a + 2] * bl[3 * l + 2];
            for (int q1 = head[l]; q1 < head[l + 1]; q1++) {
                const int k1 = lst[q1], i1 = B->hidx[B->E[k1].pose]; const double *B1 = Hpl + (size_t)k1 * 18;
                double BD[18];
                for (int a = 0; a < 6; a++) for (int c = 0; c < 3; c++) BD[3 * a + c] = B1[3 * a] * Di[c] + B1[3 * a + 1] * Di[3 + c] + B1[3 * a + 2] * Di[6 + c];
                for (int a = 0; a < 6; a++) coef[6 * i1 + a] += B1[3 * a] * db[0] + B1[3 * a + 1] * db[1] + B1[3 * a + 2] * db[2];
                for (int q2 = head[l]; q2 < head[l + 1]; q2++) {
                    const int k2 = lst[q2], i2 = B->hidx[B->E[k2].pose]; const double *B2 = Hpl + (size_t)k2 * 18;
                    if (g_ba_dense) {
                        for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++)
                            S[(size_t)(6 * i1 + a) * NP + 6 * i2 + c] -= BD[3 * a] * B2[3 * c] + BD[3 * a + 1] * B2[3 * c + 1] + BD[3 * a + 2] * B2[3 * c + 2];
                    } else if (i2 <= i1) {                         /* the factorisation reads the lower triangle only: the same entries the dense path reads */
                        for (int a = 0; a < 6; a++) for (int c = 0; c < (i2 < i1 ? 6 : a + 1); c++)
                            *env_at(M, 6 * i1 + a, 6 * i2 + c) -= BD[3 * a] * B2[3 * c] + BD[3 * a + 1] * B2[3 * c + 1] + BD[3 * a + 2] * B2[3 * c + 2];
                    }
                }
            }
        }
        free(head); free(lst); free(fill);
    }
    double *bs = (double *)malloc(sizeof(double) * (NP > 0 ? NP : 1));
    for (int i = 0; i < NP; i++) bs[i] = bp[i] - coef[i];
    if (NP > 0) ok2 = g_ba_dense ? dense_ldlt_solve(S, NP, bs, xp) : env_ldlt_solve(M, bs, xp);
    free(bs);
    if (ok2) {                                                       /* xl = Dinv (bl - Hpl^T xp) */
        double *cl = (double *)malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1) * 3); memcpy(cl, bl, sizeof(double) * (size_t)nl * 3);
        for (int k = 0; k < B->ne; k++) {
            const bedge *e = &B->E[k]; const int hp = B->hidx[e->pose];
            if (e->level != 0 || hp < 0) continue;
            const double *Bk = Hpl + (size_t)k * 18;
            for (int c = 0; c < 3; c++) { double s = 0; for (int a = 0; a < 6; a++) s += Bk[3 * a + c] * xp[6 * hp + a]; cl[3 * e->point + c] -= s; }
        }
        for (int l = 0; l < nl; l++) { const double *Di = Dinv + (size_t)l * 9; for (int a = 0; a < 3; a++) xl[3 * l + a] = Di[3 * a] * cl[3 * l] + Di[3 * a + 1] * cl[3 * l + 1] + Di[3 * a + 2] * cl[3 * l + 2]; }
        free(cl);
    }
    return ok2;
}

/* one optimizer.optimize(iterations) call on the level-0 edges; trace rows: {chi2, lambda, trials} */
static int ba_optimize(ba_t *B, int iterations, double *trace)
{
    const int nf = B->nfree, nl = B->nl, NP = 6 * nf;
    /* active vertices: a point takes part only if it has an active edge (initializeOptimization, sparse_optimizer.cpp:218-245) */
    uint8_t *pt_active = (uint8_t *)calloc(nl > 0 ? nl : 1, 1);
    for (int k = 0; k < B->ne; k++) if (B->E[k].level == 0) pt_active[B->E[k].point] = 1;
    double *Hpp = (double *)malloc(sizeof(double) * (size_t)(nf > 0 ? nf : 1) * 36), *bp = (double *)malloc(sizeof(double) * (NP > 0 ? NP : 1));
    double *Hll = (double *)malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1) * 9), *bl = (double *)malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1) * 3);
    double *Hpl = (double *)malloc(sizeof(double) * (size_t)(B->ne > 0 ? B->ne : 1) * 18);
    double *S = g_ba_dense ? (double *)malloc(sizeof(double) * (size_t)(NP > 0 ? NP : 1) * (NP > 0 ? NP : 1)) : NULL;
    envmat Senv; env_build(B, &Senv);
    double *xp = (double *)calloc(NP > 0 ? NP : 1, sizeof(double)), *xl = (double *)calloc((size_t)(nl > 0 ? nl : 1) * 3, sizeof(double));
    double *Dinv = (double *)malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1) * 9), *coef = (double *)malloc(sizeof(double) * (NP > 0 ? NP : 1));
    se3q *Tb = (se3q *)malloc(sizeof(se3q) * B->np); double *Xb = (double *)malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1) * 3);
    ba_buf W = { Hpp, bp, Hll, bl, Hpl, S, xp, xl, Dinv, coef, pt_active, Senv };
    double lambda = -1, ni = 2; int nBadLM = 0, iters = 0;
    for (int it = 0; it < iterations; it++) {
        if (B->stop && *B->stop) break;                                   /* SparseOptimizer::terminate() */
        double currentChi = ba_active_chi2(B, 1), tempChi = currentChi; const double iniChi = currentChi;
        ba_build(B, &W);
        if (it == 0) {                                                       /* computeLambdaInit over all active vertices */
            double maxd = 0;
            for (int i = 0; i < nf; i++) for (int j = 0; j < 6; j++) { const double v = fabs(Hpp[(size_t)i * 36 + 7 * j]); if (v > maxd) maxd = v; }
            for (int i = 0; i < nl; i++) if (pt_active[i]) for (int j = 0; j < 3; j++) { const double v = fabs(Hll[(size_t)i * 9 + 4 * j]); if (v > maxd) maxd = v; }
            lambda = 1e-5 * maxd; ni = 2; nBadLM = 0;
        }
        double rho = 0; int qmax = 0;
        do {
            memcpy(Tb, B->T, sizeof(se3q) * B->np); memcpy(Xb, B->X, sizeof(double) * (size_t)nl * 3);      /* push */
            const int ok2 = ba_solve(B, &W, lambda);
            /* ---- update (oplus); g2o applies _x even when the solve failed — _x then holds the previous solution */
            for (int i = 0; i < B->np; i++) { const int hp = B->hidx[i]; if (hp < 0) continue; se3q ex, up; se3_exp(xp + 6 * hp, &ex); se3_mul(&ex, &B->T[i], &up); B->T[i] = up; }
            for (int l = 0; l < nl; l++) if (pt_active[l]) for (int a = 0; a < 3; a++) B->X[3 * l + a] += xl[3 * l + a];
            tempChi = ba_active_chi2(B, 1);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0;
            for (int i = 0; i < NP; i++) scale += xp[i] * (lambda * xp[i] + bp[i]);
            for (int l = 0; l < nl; l++) if (pt_active[l]) for (int a = 0; a < 3; a++) scale += xl[3 * l + a] * (lambda * xl[3 * l + a] + bl[3 * l + a]);
            scale += 1e-3; rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3); alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                lambda *= (alpha > 1. / 3. ? alpha : 1. / 3.); ni = 2; currentChi = tempChi;
            } else { lambda *= ni; ni *= 2; memcpy(B->T, Tb, sizeof(se3q) * B->np); memcpy(B->X, Xb, sizeof(double) * (size_t)nl * 3); }
            qmax++;
        } while (rho < 0 && qmax < 10 && !(B->stop && *B->stop));
        iters = it + 1;
        if (trace) { trace[3 * it] = currentChi; trace[3 * it + 1] = lambda; trace[3 * it + 2] = qmax; }
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) break;
    }
    env_free(&Senv);
    free(pt_active); free(Hpp); free(bp); free(Hll); free(bl); free(Hpl); free(S); free(xp); free(xl); free(Dinv); free(coef); free(Tb); free(Xb);
    return iters;
}

/* flattened problem -> internal state (vertices in fp64, every edge active with the Huber kernel on) */
static void ba_setup(ba_t *Bp, int np, const float *poses, const uint8_t *pose_fixed, int nl, const float *points,
                     int ne, const int *e_pose, const int *e_point, const float *e_obs, const float *e_info,
                     float fx, float fy, float cx, float cy, float bf, const int *stop_flag)
{
    ba_t B; memset(&B, 0, sizeof B);
    B.np = np; B.nl = nl; B.ne = ne; B.cam.fx = fx; B.cam.fy = fy; B.cam.cx = cx; B.cam.cy = cy; B.cam.bf = bf;
    B.dMono = (float)sqrt(5.991); B.dStereo = (float)sqrt(7.815); B.stop = (const volatile int *)stop_flag;
    B.T = (se3q *)malloc(sizeof(se3q) * (np > 0 ? np : 1)); B.X = (double *)malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1) * 3);
    B.hidx = (int *)malloc(sizeof(int) * (np > 0 ? np : 1)); B.E = (bedge *)calloc(ne > 0 ? ne : 1, sizeof(bedge));
    for (int i = 0; i < np; i++) { se3_from_cv(poses + 16 * i, &B.T[i]); B.hidx[i] = pose_fixed[i] ? -1 : B.nfree++; }
    for (int i = 0; i < 3 * nl; i++) B.X[i] = points[i];
    for (int k = 0; k < ne; k++) {
        bedge *e = &B.E[k]; e->pose = e_pose[k]; e->point = e_point[k]; e->stereo = !(e_obs[3 * k + 2] < 0); e->level = 0; e->robust = 1;
        e->obs[0] = e_obs[3 * k]; e->obs[1] = e_obs[3 * k + 1]; e->obs[2] = e->stereo ? e_obs[3 * k + 2] : 0; e->info = e_info[k];
    }
    *Bp = B;
}

int orc_local_ba(int np, float *poses, const uint8_t *pose_fixed, int nl, float *points,
                 int ne, const int *e_pose, const int *e_point, const float *e_obs, const float *e_info,
                 float fx, float fy, float cx, float cy, float bf, const int *stop_flag,
                 uint8_t *e_erase, double *trace /* 15*3 or NULL */, int *iters_out /* 2 or NULL */)
{
    ba_t B;
    ba_setup(&B, np, poses, pose_fixed, nl, points, ne, e_pose, e_point, e_obs, e_info, fx, fy, cx, cy, bf, stop_flag);
    memset(e_erase, 0, ne);
    if (stop_flag && *stop_flag) goto done;                                   /* Optimizer.cc:655-657 */
    { const int it1 = ba_optimize(&B, 5, trace); if (iters_out) iters_out[0] = it1; }
    if (!(stop_flag && *stop_flag)) {                                         /* :662-707 */
        for (int k = 0; k < ne; k++) {
            bedge *e = &B.E[k];
            double p[3]; se3_map(&B.T[e->pose], &B.X[3 * e->point], p);
            if (bedge_chi2(e) > (e->stereo ? 7.815 : 5.991) || !(p[2] > 0.0)) e->level = 1;
            e->robust = 0;
        }
        const int it2 = ba_optimize(&B, 10, trace ? trace + 45 : NULL); if (iters_out) iters_out[1] = it2;
    }
    for (int k = 0; k < ne; k++) {                                            /* :709-742 */
        bedge *e = &B.E[k];
        double p[3]; se3_map(&B.T[e->pose], &B.X[3 * e->point], p);
        if (bedge_chi2(e) > (e->stereo ? 7.815 : 5.991) || !(p[2] > 0.0)) e_erase[k] = 1;
    }
    for (int i = 0; i < np; i++) if (pose_fixed[i] != 1) se3_to_cv(&B.T[i], poses + 16 * i);   /* :762-768: every LOCAL KF is rewritten (pose_fixed 2 = local KF with mnId 0: fixed vertex, still rewritten) */
    for (int i = 0; i < 3 * nl; i++) points[i] = (float)B.X[i];                                  /* :771-777 */
done:
    free(B.T); free(B.X); free(B.hidx); free(B.E);
    return 0;
}

/* Optimizer::BundleAdjustment (Optimizer.cc:49-237) on the flattened graph: every keyframe is a vertex (fixed iff mnId == 0: pose_fixed != 0), every map point with
 * at least one edge is a marginalised vertex, ONE optimize(nIterations) over all edges (level 0), Huber deltas sqrt(5.99) / sqrt(7.815) as floats when bRobust
 * (:83-84; note 5.99, not LocalBA's 5.991), no outlier classification.  Every keyframe pose is rewritten through SE3Quat (:200-214, the fixed one included), every
 * included point through Vector3d -> float (:216-236).  Points without edges are "not included" (vbNotIncludedMP): untouched. */
int orc_bundle_adjustment(int np, float *poses, const uint8_t *pose_fixed, int nl, float *points,
                          int ne, const int *e_pose, const int *e_point, const float *e_obs, const float *e_info,
                          float fx, float fy, float cx, float cy, float bf, int n_iterations, int robust, const int *stop_flag,
                          double *trace /* n_iterations*3 or NULL */, int *iters_out /* 1 or NULL */)
{
    ba_t B;
    ba_setup(&B, np, poses, pose_fixed, nl, points, ne, e_pose, e_point, e_obs, e_info, fx, fy, cx, cy, bf, stop_flag);
    B.dMono = (float)sqrt(5.99); B.dStereo = (float)sqrt(7.815);
    for (int k = 0; k < ne; k++) B.E[k].robust = robust ? 1 : 0;
    uint8_t *included = (uint8_t *)calloc(nl > 0 ? nl : 1, 1);
    for (int k = 0; k < ne; k++) included[e_point[k]] = 1;
    const int it = ba_optimize(&B, n_iterations, trace);
    if (iters_out) iters_out[0] = it;
    for (int i = 0; i < np; i++) se3_to_cv(&B.T[i], poses + 16 * i);
    for (int i = 0; i < nl; i++) if (included[i]) for (int c = 0; c < 3; c++) points[3 * i + c] = (float)B.X[3 * i + c];
    free(included);
    free(B.T); free(B.X); free(B.hidx); free(B.E);
    return 0;
}

/* ---- known-answer taps (tests/test_oracle_kat.py): the pieces above exposed one at a time, so that the tests can pin them against independent
 * arithmetic (central differences for the Jacobians, a dense solve of the full normal equations for the Schur path) ---- */
int orc_kat_ba_edge(const float *Tcw, const double *X, const double *obs, int stereo, double fx, double fy, double cx, double cy, double bf,
                    const double *dpose, const double *dpoint, double *err, double *Jpose, double *Jpoint)
{
    se3q T, ex, Tp; se3_from_cv(Tcw, &T); se3_exp(dpose, &ex); se3_mul(&ex, &T, &Tp);             /* VertexSE3Expmap::oplusImpl */
    const double Xp[3] = { X[0] + dpoint[0], X[1] + dpoint[1], X[2] + dpoint[2] };                 /* VertexSBAPointXYZ::oplusImpl */
    bcam cam = { fx, fy, cx, cy, bf };
    bedge e; memset(&e, 0, sizeof e); e.stereo = stereo; e.obs[0] = obs[0]; e.obs[1] = obs[1]; e.obs[2] = obs[2]; e.info = 1;
    bedge_error(&e, &Tp, Xp, &cam);
    double A[3][3], Bj[3][6]; bedge_jacobians(stereo, &Tp, Xp, &cam, A, Bj);
    for (int d = 0; d < 3; d++) { err[d] = e.err[d]; for (int c = 0; c < 3; c++) Jpoint[3 * d + c] = A[d][c]; for (int c = 0; c < 6; c++) Jpose[6 * d + c] = Bj[d][c]; }
    return 0;
}

int orc_kat_ba_step(int np, const float *poses, const uint8_t *pose_fixed, int nl, const float *points,
                    int ne, const int *e_pose, const int *e_point, const float *e_obs, const float *e_info,
                    float fx, float fy, float cx, float cy, float bf, double lambda, int robust,
                    double *Hpp, double *bp, double *Hll, double *bl, double *Hpl, double *xp, double *xl, int *hidx_out)
{
    ba_t B;
    ba_setup(&B, np, poses, pose_fixed, nl, points, ne, e_pose, e_point, e_obs, e_info, fx, fy, cx, cy, bf, NULL);
    for (int k = 0; k < ne; k++) B.E[k].robust = robust;
    const int nf = B.nfree, NP = 6 * nf;
    ba_buf W; memset(&W, 0, sizeof W);
    W.pt_active = (uint8_t *)calloc(nl > 0 ? nl : 1, 1);
    for (int k = 0; k < ne; k++) W.pt_active[B.E[k].point] = 1;
    W.Hpp = Hpp; W.bp = bp; W.Hll = Hll; W.bl = bl; W.Hpl = Hpl; W.xp = xp; W.xl = xl;
    W.S = g_ba_dense ? (double *)malloc(sizeof(double) * (size_t)(NP > 0 ? NP : 1) * (NP > 0 ? NP : 1)) : NULL;
    env_build(&B, &W.Senv);
    W.Dinv = (double *)malloc(sizeof(double) * (size_t)(nl > 0 ? nl : 1) * 9); W.coef = (double *)malloc(sizeof(double) * (NP > 0 ? NP : 1));
    (void)ba_active_chi2(&B, 1);                                              /* computeActiveErrors */
    ba_build(&B, &W);
    const int ok = ba_solve(&B, &W, lambda);
    for (int i = 0; i < np; i++) hidx_out[i] = B.hidx[i];
    free(W.pt_active); free(W.S); free(W.Dinv); free(W.coef); env_free(&W.Senv);
    free(B.T); free(B.X); free(B.hidx); free(B.E);
    return ok;
}
