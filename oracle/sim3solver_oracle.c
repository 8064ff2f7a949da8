/* sim3solver_oracle.c — TEST INFRASTRUCTURE ONLY (never linked, imported or executed by the product path).
 *
 * CPU restatement of Sim3Solver (src/sg-slam/src/Sim3Solver.cc), the RANSAC initialiser of LoopClosing::ComputeSim3 (LoopClosing.cc:274-301):
 *   SetRansacParameters :113-138, iterate :140-208, ComputeCentroid :216-225, ComputeSim3 (Horn 1987) :228-337, CheckInliers :340-365, Project :383-405,
 *   FromCameraToImage :407-425; DUtils::Random::RandomInt (Thirdparty/DBoW2/DUtils/Random.cpp:71-74).
 * The cv::Mat arithmetic is restated operation by operation from OpenCV 3.4 (cv::reduce, MatExpr scaling = convertTo with the float of the factor, cv::gemm small / generic
 * paths, Mat::dot, cv::eigen -> Jacobi for a symmetric float matrix, cv::norm, cv::Rodrigues in double).  OpenCV is absent here: parity unpinned.
 *
 * The constructor's flattening (:40-111: camera-frame points X3Dc = Rcw * Xw + tcw, the 9.210 * sigma^2 error bounds, index bookkeeping) stays with the caller: the
 * functions take the N compacted correspondences.  Random draws: the reference calls the process-global rand(); here the caller passes the raw rand() values
 * (3 per iteration), or uses orc_glibc_rand below — glibc's TYPE_3 generator, srand(seed) then rand() — to produce them.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int N, fix_scale;
    float *X1, *X2, *P1im1, *P2im2, *maxErr1, *maxErr2;     /* N x 3, N x 3, N x 2, N x 2, N, N */
    float K1[4], K2[4];                                      /* fx, fy, cx, cy */
    double prob; int minInliers, maxIts, nIterations, nBestInliers;
    uint8_t *inl_i, *best_inl;
    float R12i[9], t12i[3], s12i, T12i[16], T21i[16];
    float bestT12[16], bestR[9], bestt[3], bests;
} orc_s3;

/* glibc rand(): TYPE_3 additive feedback generator (r[i] = r[i-3] + r[i-31], output >> 1), seeded as srand() does (seed 0 -> 1, 16807 LCG, 310 discarded values) */
typedef struct { int32_t r[34]; int f, b; } orc_grand;
void orc_glibc_srand(orc_grand *g, unsigned seed)
{
    int32_t word = seed ? (int32_t)seed : 1;
    g->r[0] = word;
    for (int i = 1; i < 31; i++) {
        const long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w; g->r[i] = word;
    }
    g->f = 3; g->b = 0;
    for (int i = 0; i < 310; i++) { g->r[g->f] = (int32_t)((uint32_t)g->r[g->f] + (uint32_t)g->r[g->b]); g->f = (g->f + 1) % 31; g->b = (g->b + 1) % 31; }
}
int orc_glibc_rand(orc_grand *g)
{
    g->r[g->f] = (int32_t)((uint32_t)g->r[g->f] + (uint32_t)g->r[g->b]);
    const int32_t out = (int32_t)(((uint32_t)g->r[g->f]) >> 1);
    g->f = (g->f + 1) % 31; g->b = (g->b + 1) % 31;
    return out;
}
void orc_glibc_rand_sequence(unsigned seed, int n, int32_t *out) { orc_grand g; orc_glibc_srand(&g, seed); for (int i = 0; i < n; i++) out[i] = orc_glibc_rand(&g); }

static float gemm_small3(const float *arow, const float *b, int bstep, double alpha, double beta, float c)
{   /* cv::gemm hand-unrolled path for len 3: float dot left to right, then (float)(t * alpha + c * beta) */
    const float t = arow[0] * b[0] + arow[1] * b[bstep] + arow[2] * b[2 * bstep];
    return (float)(t * alpha + c * beta);
}

/* FromCameraToImage :407-425 */
static void to_image(int n, const float *X, const float *K, float *out)
{
    for (int i = 0; i < n; i++) {
        const float invz = 1 / X[3 * i + 2], x = X[3 * i] * invz, y = X[3 * i + 1] * invz;
        out[2 * i] = K[0] * x + K[2]; out[2 * i + 1] = K[1] * y + K[3];
    }
}

void orc_s3_set_ransac_parameters(orc_s3 *s, double probability, int minInliers, int maxIterations);
orc_s3 *orc_s3_create(int N, const float *X3Dc1, const float *X3Dc2, const float *maxErr1, const float *maxErr2, const float *K1, const float *K2, int fix_scale)
{
    orc_s3 *s = (orc_s3 *)calloc(1, sizeof(orc_s3));
    const size_t n = (size_t)(N > 0 ? N : 1);
    s->N = N; s->fix_scale = fix_scale;
    s->X1 = (float *)malloc(12 * n); s->X2 = (float *)malloc(12 * n); s->P1im1 = (float *)malloc(8 * n); s->P2im2 = (float *)malloc(8 * n);
    s->maxErr1 = (float *)malloc(4 * n); s->maxErr2 = (float *)malloc(4 * n); s->inl_i = (uint8_t *)calloc(n, 1); s->best_inl = (uint8_t *)calloc(n, 1);
    memcpy(s->X1, X3Dc1, 12 * (size_t)N); memcpy(s->X2, X3Dc2, 12 * (size_t)N); memcpy(s->maxErr1, maxErr1, 4 * (size_t)N); memcpy(s->maxErr2, maxErr2, 4 * (size_t)N);
    memcpy(s->K1, K1, 16); memcpy(s->K2, K2, 16);
    to_image(N, s->X1, s->K1, s->P1im1); to_image(N, s->X2, s->K2, s->P2im2);
    orc_s3_set_ransac_parameters(s, 0.99, 6, 300);                     /* the constructor ends with SetRansacParameters() (:110, defaults Sim3Solver.h:41) */
    return s;
}
void orc_s3_destroy(orc_s3 *s) { if (!s) return; free(s->X1); free(s->X2); free(s->P1im1); free(s->P2im2); free(s->maxErr1); free(s->maxErr2); free(s->inl_i); free(s->best_inl); free(s); }

/* SetRansacParameters :113-138 */
void orc_s3_set_ransac_parameters(orc_s3 *s, double probability, int minInliers, int maxIterations)
{
    s->prob = probability; s->minInliers = minInliers; s->maxIts = maxIterations;
    const float epsilon = (float)s->minInliers / s->N;
    int nIterations;
    if (s->minInliers == s->N) nIterations = 1;
    else nIterations = (int)ceil(log(1 - s->prob) / log(1 - pow(epsilon, 3)));
    const int m = nIterations < s->maxIts ? nIterations : s->maxIts;
    s->maxIts = m > 1 ? m : 1;
    s->nIterations = 0;
}
int orc_s3_max_iterations(const orc_s3 *s) { return s->maxIts; }

/* cv::eigen on a symmetric float matrix = JacobiImpl_<float> (OpenCV lapack.cpp): eigenvalues descending, eigenvectors as rows */
static void jacobi4(float *A /* 4x4, destroyed */, float *W, float *V)
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
        const float y = (float)((W[l] - W[k]) * 0.5);
        float t = fabsf(y) + hypotf(p, y);
        float s = hypotf(p, t);
        const float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0) s = -s, t = -t;
        A[n * k + l] = 0;
        W[k] -= t; W[l] += t;
        float a0, b0;
#define ROT(v0, v1) a0 = v0, b0 = v1, v0 = a0 * c - b0 * s, v1 = a0 * s + b0 * c
        for (i = 0; i < k; i++) ROT(A[n * i + k], A[n * i + l]);
        for (i = k + 1; i < l; i++) ROT(A[n * k + i], A[n * i + l]);
        for (i = l + 1; i < n; i++) ROT(A[n * k + i], A[n * l + i]);
        for (i = 0; i < n; i++) ROT(V[n * k + i], V[n * l + i]);
#undef ROT
        for (int j = 0; j < 2; j++) {
            const int idx = j == 0 ? k : l;
            if (idx < n - 1) { for (m = idx + 1, mv = fabsf(A[n * idx + m]), i = idx + 2; i < n; i++) { const float val = fabsf(A[n * idx + i]); if (mv < val) mv = val, m = i; } indR[idx] = m; }
            if (idx > 0) { for (m = 0, mv = fabsf(A[idx]), i = 1; i < idx; i++) { const float val = fabsf(A[n * i + idx]); if (mv < val) mv = val, m = i; } indC[idx] = m; }
        }
    }
    for (k = 0; k < n - 1; k++) {
        m = k;
        for (i = k + 1; i < n; i++) if (W[m] < W[i]) m = i;
        if (k != m) { float tw = W[m]; W[m] = W[k]; W[k] = tw; for (i = 0; i < n; i++) { const float tv = V[n * m + i]; V[n * m + i] = V[n * k + i]; V[n * k + i] = tv; } }
    }
}

/* cv::Rodrigues(vector -> matrix): computed in double, stored as float */
static void rodrigues(const float *v, float *R)
{
    const double rx0 = v[0], ry0 = v[1], rz0 = v[2];
    const double theta = sqrt(rx0 * rx0 + ry0 * ry0 + rz0 * rz0);
    if (theta < DBL_EPSILON) { for (int i = 0; i < 9; i++) R[i] = (i % 4) == 0 ? 1.f : 0.f; return; }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    const double rx = rx0 * itheta, ry = ry0 * itheta, rz = rz0 * itheta;
    const double rrt[9] = { rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz };
    const double rx_[9] = { 0, -rz, ry, rz, 0, -rx, -ry, rx, 0 };
    for (int i = 0; i < 9; i++) R[i] = (float)(c * ((i % 4) == 0 ? 1. : 0.) + c1 * rrt[i] + s * rx_[i]);
}

/* Mat::dot without SIMD (dotProd_ template): double accumulation, four products per step */
static double dot_f32(const float *a, const float *b, int len)
{
    double r = 0; int i = 0;
    for (; i <= len - 4; i += 4) r += (double)a[i] * b[i] + (double)a[i + 1] * b[i + 1] + (double)a[i + 2] * b[i + 2] + (double)a[i + 3] * b[i + 3];
    for (; i < len; i++) r += (double)a[i] * b[i];
    return r;
}

/* ComputeSim3 :228-337 on the 3 x 3 matrices P1, P2 (column i = point i, row-major storage) */
static void compute_sim3(orc_s3 *s, const float *P1, const float *P2)
{
    float Pr1[9], Pr2[9], O1[3], O2[3];
    const float third = (float)(1.0 / 3);                       /* C = C / P.cols : Mat / double -> convertTo by the float of 1/3 */
    for (int r = 0; r < 3; r++) {
        O1[r] = (P1[3 * r] + P1[3 * r + 1] + P1[3 * r + 2]) * third;          /* cv::reduce SUM in float, left to right */
        O2[r] = (P2[3 * r] + P2[3 * r + 1] + P2[3 * r + 2]) * third;
        for (int c = 0; c < 3; c++) { Pr1[3 * r + c] = P1[3 * r + c] - O1[r]; Pr2[3 * r + c] = P2[3 * r + c] - O2[r]; }
    }
    float M[9];                                                  /* Pr2 * Pr1.t(): gemm with a transpose flag -> generic path, double accumulation */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += (double)Pr2[3 * i + k] * (double)Pr1[3 * j + k]; M[3 * i + j] = (float)a; }
    const double N11 = M[0] + M[4] + M[8], N12 = M[5] - M[7], N13 = M[6] - M[2], N14 = M[1] - M[3];
    const double N22 = M[0] - M[4] - M[8], N23 = M[1] + M[3], N24 = M[6] + M[2];
    const double N33 = -M[0] + M[4] - M[8], N34 = M[5] + M[7], N44 = -M[0] - M[4] + M[8];
    float Nm[16] = { (float)N11, (float)N12, (float)N13, (float)N14, (float)N12, (float)N22, (float)N23, (float)N24,
                     (float)N13, (float)N23, (float)N33, (float)N34, (float)N14, (float)N24, (float)N34, (float)N44 };
    float eval[4], evec[16];
    jacobi4(Nm, eval, evec);
    float vec[3] = { evec[1], evec[2], evec[3] };
    const double nv = sqrt((double)vec[0] * vec[0] + (double)vec[1] * vec[1] + (double)vec[2] * vec[2]);         /* cv::norm */
    const double ang = atan2(nv, (double)evec[0]);
    const float sc = (float)((2 * ang) * (1. / nv));            /* vec = 2*ang*vec/norm(vec): one MatExpr scale */
    for (int i = 0; i < 3; i++) vec[i] = vec[i] * sc;
    rodrigues(vec, s->R12i);
    float P3[9];                                                 /* mR12i * Pr2: 3x3 * 3x3, hand-unrolled path */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) P3[3 * i + j] = gemm_small3(s->R12i + 3 * i, Pr2 + j, 3, 1.0, 0.0, 0.f);
    if (!s->fix_scale) {
        const double nom = dot_f32(Pr1, P3, 9);
        double den = 0;
        for (int i = 0; i < 9; i++) { const float sq = P3[i] * P3[i]; den += sq; }      /* cv::pow(P3, 2): x * x in float */
        s->s12i = (float)(nom / den);
    } else s->s12i = 1.0f;
    /* mt12i = O1 - ms12i*mR12i*O2 : one gemm(R, O2, -s, O1, 1) */
    for (int i = 0; i < 3; i++) s->t12i[i] = gemm_small3(s->R12i + 3 * i, O2, 1, -(double)s->s12i, 1.0, O1[i]);
    memset(s->T12i, 0, sizeof s->T12i); memset(s->T21i, 0, sizeof s->T21i); s->T12i[15] = 1.f; s->T21i[15] = 1.f;
    float sRinv[9];
    const float inv_s = (float)(1.0 / (double)s->s12i);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { s->T12i[4 * i + j] = s->R12i[3 * i + j] * s->s12i; sRinv[3 * i + j] = s->R12i[3 * j + i] * inv_s; s->T21i[4 * i + j] = sRinv[3 * i + j]; }
    for (int i = 0; i < 3; i++) { s->T12i[4 * i + 3] = s->t12i[i]; s->T21i[4 * i + 3] = gemm_small3(sRinv + 3 * i, s->t12i, 1, -1.0, 0.0, 0.f); }
}

/* Project :383-405 + CheckInliers :340-365 */
static int check_inliers(orc_s3 *s)
{
    int n = 0;
    for (int i = 0; i < s->N; i++) {
        float p21[2], p12[2];
        for (int dir = 0; dir < 2; dir++) {
            const float *T = dir == 0 ? s->T12i : s->T21i, *X = dir == 0 ? s->X2 + 3 * i : s->X1 + 3 * i, *K = dir == 0 ? s->K1 : s->K2;
            const float pc[3] = { gemm_small3(T, X, 1, 1.0, 1.0, T[3]), gemm_small3(T + 4, X, 1, 1.0, 1.0, T[7]), gemm_small3(T + 8, X, 1, 1.0, 1.0, T[11]) };
            const float invz = 1 / pc[2], x = pc[0] * invz, y = pc[1] * invz;
            float *o = dir == 0 ? p21 : p12;
            o[0] = K[0] * x + K[2]; o[1] = K[1] * y + K[3];
        }
        const float d1[2] = { s->P1im1[2 * i] - p21[0], s->P1im1[2 * i + 1] - p21[1] }, d2[2] = { p12[0] - s->P2im2[2 * i], p12[1] - s->P2im2[2 * i + 1] };
        const float err1 = (float)((double)d1[0] * d1[0] + (double)d1[1] * d1[1]), err2 = (float)((double)d2[0] * d2[0] + (double)d2[1] * d2[1]);
        s->inl_i[i] = err1 < s->maxErr1[i] && err2 < s->maxErr2[i];
        n += s->inl_i[i];
    }
    return n;
}

/* iterate :140-208.  rand_draws: 3 raw rand() values per iteration (only those of the iterations actually run are consumed).  Returns 1 when a model with more than
 * minInliers inliers is found (T12 = mBestT12, inliers[N], *nInliers), 0 otherwise; *noMore as bNoMore; *iters_run = iterations executed by this call. */
int orc_s3_iterate(orc_s3 *s, int nIterations, const int32_t *rand_draws, float *T12, int *noMore, uint8_t *inliers, int *nInliers, int *iters_run)
{
    *noMore = 0; *nInliers = 0; *iters_run = 0;
    memset(inliers, 0, (size_t)(s->N > 0 ? s->N : 1));
    if (s->N < s->minInliers) { *noMore = 1; return 0; }
    int *avail = (int *)malloc(sizeof(int) * (size_t)(s->N > 0 ? s->N : 1));
    int cur = 0;
    while (s->nIterations < s->maxIts && cur < nIterations) {
        cur++; s->nIterations++;
        int na = s->N; for (int i = 0; i < na; i++) avail[i] = i;
        float P1[9], P2[9];
        for (int i = 0; i < 3; i++) {
            const int d = (na - 1) - 0 + 1;                                                                       /* RandomInt(0, size - 1), Random.cpp:71-74 */
            const int randi = (int)(((double)rand_draws[3 * (cur - 1) + i] / ((double)2147483647 + 1.0)) * d) + 0;
            const int idx = avail[randi];
            for (int r = 0; r < 3; r++) { P1[3 * r + i] = s->X1[3 * idx + r]; P2[3 * r + i] = s->X2[3 * idx + r]; }
            avail[randi] = avail[na - 1]; na--;
        }
        compute_sim3(s, P1, P2);
        const int ninl = check_inliers(s);
        if (ninl >= s->nBestInliers) {
            memcpy(s->best_inl, s->inl_i, (size_t)s->N); s->nBestInliers = ninl;
            memcpy(s->bestT12, s->T12i, sizeof s->bestT12); memcpy(s->bestR, s->R12i, sizeof s->bestR); memcpy(s->bestt, s->t12i, sizeof s->bestt); s->bests = s->s12i;
            if (ninl > s->minInliers) {
                *nInliers = ninl; memcpy(inliers, s->inl_i, (size_t)s->N); memcpy(T12, s->bestT12, sizeof s->bestT12);
                *iters_run = cur; free(avail); return 1;
            }
        }
    }
    *iters_run = cur;
    if (s->nIterations >= s->maxIts) *noMore = 1;
    free(avail);
    return 0;
}
void orc_s3_best(const orc_s3 *s, float *R, float *t, float *scale) { memcpy(R, s->bestR, 36); memcpy(t, s->bestt, 12); *scale = s->bests; }

/* test tap: Horn's closed form on one triple */
void orc_s3_compute(const float *P1, const float *P2, int fix_scale, float *R, float *t, float *scale)
{
    orc_s3 s; memset(&s, 0, sizeof s); s.fix_scale = fix_scale;
    compute_sim3(&s, P1, P2);
    memcpy(R, s.R12i, 36); memcpy(t, s.t12i, 12); *scale = s.s12i;
}
