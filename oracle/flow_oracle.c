/* oracle/flow_oracle.c — CPU ORACLE for the inputs of the dynamic-feature mask (tier N1).  TEST INFRASTRUCTURE ONLY.
 *
 * Restates what Frame::RmDynamicPointWithSemanticAndGeometry does before its erase loop
 *   /root/reference/src/sg-slam/src/Frame.cc:430-472
 *     :445      cv::calcOpticalFlowPyrLK(imGray, imGrayPre, Curpoint, Prepoint, State, Err, Size(21,21), 3,
 *                                        TermCriteria(ITER|EPS, 30, 0.01))           (current -> previous frame, every keypoint)
 *     :454-467  selection of the pairs whose PREVIOUS position lies outside the previous frame's person boxes
 *     :469-472  cv::findFundamentalMat(cur, prev, FM_RANSAC, 1.0, 0.99) on the selected pairs (> 20) or on all pairs
 * Both OpenCV 3.4.15 functions are third-party code that is neither vendored in the reference tree nor installed here
 * (video/src/lkpyramid.cpp, imgproc/src/pyramids.cpp, calib3d/src/fundam.cpp, calib3d/src/ptsetreg.cpp,
 * core/src/mathfuncs.cpp solveCubic, core RNG); they are restated from the published 3.4 sources:
 *      ==> PARITY UNPINNED at the OpenCV boundary <==        (no golden vectors exist; see DESIGN.md "Oracle")
 *
 * What is exact integer arithmetic in OpenCV (pyrDown, Scharr derivatives, the bilinear window samples, the
 * products summed into the 2x2 gradient matrix and the mismatch vector) is exact here.  The SUMS are accumulated in
 * OpenCV in a type that depends on the build (lkpyramid.cpp: `typedef float acctype` on x86 — there additionally in
 * an SSE2 lane order — and `typedef int64 acctype` on ARM without NEON); `acc_mode` selects
 *      0  float, scalar C++ loop order (row-major over the window)
 *      1  int64 (exact), converted to float once        <- the order-free variant the device implements bit-exactly
 * Tracked positions of the two variants differ by float rounding noise only (tests state the tolerance).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  Compile with -ffp-contract=off.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
/* cv::borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

/* ------------------------------------------------------------------------------------------------------------------
 * cv::pyrDown(src, dst, Size((w+1)/2,(h+1)/2)) for CV_8U, BORDER_DEFAULT (= REFLECT_101)   — imgproc/src/pyramids.cpp
 * pyrDown_<FixPtCast<uchar, 8>>: horizontal [1 4 6 4 1] into int rows, vertical the same, (sum + 128) >> 8.
 * ---------------------------------------------------------------------------------------------------------------- */
void orc_pyrdown_u8(const uint8_t *src, int sw, int sh, int spitch, uint8_t *dst, int dw, int dh, int dpitch)
{
    int *rows = (int *)malloc(sizeof(int) * 5 * (size_t)dw);
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            const uint8_t *s = src + (size_t)reflect101(2 * y - 2 + k, sh) * spitch;
            int *r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                const int x0 = reflect101(2 * x - 2, sw), x1 = reflect101(2 * x - 1, sw), x2 = reflect101(2 * x, sw),
                          x3 = reflect101(2 * x + 1, sw), x4 = reflect101(2 * x + 2, sw);
                r[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        const int *r0 = rows, *r1 = rows + dw, *r2 = rows + 2 * dw, *r3 = rows + 3 * dw, *r4 = rows + 4 * dw;
        for (int x = 0; x < dw; x++)
            dst[(size_t)y * dpitch + x] = (uint8_t)((r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x] + 128) >> 8);
    }
    free(rows);
}

/* calcSharrDeriv (lkpyramid.cpp): d[2x] = dI/dx = [-1 0 1] (x) [3 10 3]^T, d[2x+1] = dI/dy = [3 10 3] (x) [-1 0 1]^T, no scaling;
 * rows / columns outside the image are the REFLECT_101 neighbours (y-1 -> 1, y+1 -> rows-2). */
void orc_scharr_deriv(const uint8_t *src, int w, int h, int pitch, int16_t *dst /* h x w x 2 */)
{
    int *t0 = (int *)malloc(sizeof(int) * (size_t)(w + 2) * 2), *t1 = t0 + (w + 2);
    for (int y = 0; y < h; y++) {
        const uint8_t *s0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * pitch;
        const uint8_t *s1 = src + (size_t)y * pitch;
        const uint8_t *s2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * pitch;
        int *a = t0 + 1, *b = t1 + 1;
        for (int x = 0; x < w; x++) { a[x] = (int16_t)((s0[x] + s2[x]) * 3 + s1[x] * 10); b[x] = (int16_t)(s2[x] - s0[x]); }
        const int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
        a[-1] = a[x0]; a[w] = a[x1]; b[-1] = b[x0]; b[w] = b[x1];
        int16_t *d = dst + (size_t)y * w * 2;
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(a[x + 1] - a[x - 1]);
            d[2 * x + 1] = (int16_t)((b[x + 1] + b[x - 1]) * 3 + b[x] * 10);
        }
    }
    free(t0);
}

/* ------------------------------------------------------------------------------------------------------------------
 * cv::calcOpticalFlowPyrLK(prevImg = I, nextImg = J, prevPts, nextPts, status, err, winSize = (win, win), maxLevel,
 *                          TermCriteria(COUNT|EPS, max_count, epsilon), flags = 0, minEigThreshold = 1e-4)
 * SparsePyrLKOpticalFlowImpl::calc + buildOpticalFlowPyramid(withDerivatives = false) + LKTrackerInvoker::operator()
 * (video/src/lkpyramid.cpp).  `err` is not computed (the reference ignores State and Err, Frame.cc:445); status follows
 * the library's rules anyway so that tests can look at it.
 * ---------------------------------------------------------------------------------------------------------------- */
#define LK_MAXLEV 8
typedef struct { int w, h; uint8_t *img; int16_t *der; } lk_level;

static inline int lk_pix(const lk_level *L, int x, int y) { return L->img[(size_t)reflect101(y, L->h) * L->w + reflect101(x, L->w)]; }   /* copyMakeBorder(REFLECT_101) of winSize */
static inline void lk_der(const lk_level *L, int x, int y, int *dx, int *dy)                                                               /* copyMakeBorder(CONSTANT 0) */
{
    if (x < 0 || y < 0 || x >= L->w || y >= L->h) { *dx = 0; *dy = 0; return; }
    const int16_t *d = L->der + ((size_t)y * L->w + x) * 2; *dx = d[0]; *dy = d[1];
}
#define CV_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

static int lk_build(const uint8_t *img, int w, int h, int pitch, int win, int max_level, lk_level *L, int with_deriv)
{
    int nl = 0;
    L[0].w = w; L[0].h = h; L[0].img = (uint8_t *)malloc((size_t)w * h); L[0].der = NULL;
    for (int y = 0; y < h; y++) memcpy(L[0].img + (size_t)y * w, img + (size_t)y * pitch, (size_t)w);
    int level = 0;
    for (;; level++) {                                       /* buildOpticalFlowPyramid's loop */
        nl = level + 1;
        if (level == max_level) break;
        const int nw = (L[level].w + 1) / 2, nh = (L[level].h + 1) / 2;
        if (nw <= win || nh <= win) break;                   /* "return level" */
        L[level + 1].w = nw; L[level + 1].h = nh; L[level + 1].der = NULL;
        L[level + 1].img = (uint8_t *)malloc((size_t)nw * nh);
        orc_pyrdown_u8(L[level].img, L[level].w, L[level].h, L[level].w, L[level + 1].img, nw, nh, nw);
    }
    if (with_deriv)
        for (int l = 0; l < nl; l++) { L[l].der = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)L[l].w * L[l].h); orc_scharr_deriv(L[l].img, L[l].w, L[l].h, L[l].w, L[l].der); }
    return nl;
}
static void lk_free(lk_level *L, int nl) { for (int l = 0; l < nl; l++) { free(L[l].img); free(L[l].der); } }

/* returns the number of pyramid levels used.  iters (optional, n x nlevels int32): LK iterations run per point and level (diagnostic). */
int orc_lk_pyr(const uint8_t *I0, const uint8_t *J0, int w, int h, int pitch, const float *prev_pts, int n, float *next_pts, uint8_t *status,
               int win, int max_level, int max_count, double epsilon, int acc_mode, int32_t *iters)
{
    if (max_level >= LK_MAXLEV) max_level = LK_MAXLEV - 1;
    lk_level LI[LK_MAXLEV], LJ[LK_MAXLEV];
    const int nl = lk_build(I0, w, h, pitch, win, max_level, LI, 1);
    lk_build(J0, w, h, pitch, win, max_level, LJ, 0);
    max_count = max_count < 0 ? 0 : max_count > 100 ? 100 : max_count;
    epsilon = epsilon < 0. ? 0. : epsilon > 10. ? 10. : epsilon;
    epsilon *= epsilon;
    const float min_eig_threshold = (float)1e-4;
    const float half = (win - 1) * 0.5f;
    short *Iwin = (short *)malloc(sizeof(short) * 3 * (size_t)win * win), *dIwin = Iwin + (size_t)win * win;
    for (int i = 0; i < n; i++) status[i] = 1;
    if (iters) memset(iters, 0, sizeof(int32_t) * (size_t)n * nl);

    for (int level = nl - 1; level >= 0; level--) {
        const lk_level *I = &LI[level], *J = &LJ[level];
        for (int p = 0; p < n; p++) {
            float prevx = prev_pts[2 * p] * (float)(1. / (1 << level)), prevy = prev_pts[2 * p + 1] * (float)(1. / (1 << level));
            float nextx, nexty;
            if (level == nl - 1) { nextx = prevx; nexty = prevy; }
            else { nextx = next_pts[2 * p] * 2.f; nexty = next_pts[2 * p + 1] * 2.f; }
            next_pts[2 * p] = nextx; next_pts[2 * p + 1] = nexty;
            prevx -= half; prevy -= half;
            const int ipx = cv_floor_f(prevx), ipy = cv_floor_f(prevy);
            if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) { if (level == 0) status[p] = 0; continue; }
            float a = prevx - ipx, b = prevy - ipy;
            const int W_BITS = 14, W_BITS1 = 14;
            const float FLT_SCALE = 1.f / (1 << 20);
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float fA11 = 0, fA12 = 0, fA22 = 0; int64_t qA11 = 0, qA12 = 0, qA22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    const int gx = ipx + x, gy = ipy + y;
                    const int ival = CV_DESCALE(lk_pix(I, gx, gy) * iw00 + lk_pix(I, gx + 1, gy) * iw01 + lk_pix(I, gx, gy + 1) * iw10 + lk_pix(I, gx + 1, gy + 1) * iw11, W_BITS1 - 5);
                    int x00, y00, x01, y01, x10, y10, x11, y11;
                    lk_der(I, gx, gy, &x00, &y00); lk_der(I, gx + 1, gy, &x01, &y01); lk_der(I, gx, gy + 1, &x10, &y10); lk_der(I, gx + 1, gy + 1, &x11, &y11);
                    const int ixval = CV_DESCALE(x00 * iw00 + x01 * iw01 + x10 * iw10 + x11 * iw11, W_BITS1);
                    const int iyval = CV_DESCALE(y00 * iw00 + y01 * iw01 + y10 * iw10 + y11 * iw11, W_BITS1);
                    Iwin[y * win + x] = (short)ival; dIwin[(y * win + x) * 2] = (short)ixval; dIwin[(y * win + x) * 2 + 1] = (short)iyval;
                    fA11 += (float)(ixval * ixval); fA12 += (float)(ixval * iyval); fA22 += (float)(iyval * iyval);
                    qA11 += (int64_t)ixval * ixval; qA12 += (int64_t)ixval * iyval; qA22 += (int64_t)iyval * iyval;
                }
            float A11, A12, A22;
            if (acc_mode == 0) { A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE; }
            else { A11 = (float)qA11 * FLT_SCALE; A12 = (float)qA12 * FLT_SCALE; A22 = (float)qA22 * FLT_SCALE; }
            float D = A11 * A22 - A12 * A12;
            const float min_eig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
            if (min_eig < min_eig_threshold || D < FLT_EPSILON) { if (level == 0) status[p] = 0; continue; }
            D = 1.f / D;
            nextx -= half; nexty -= half;
            float pdx = 0.f, pdy = 0.f;
            int j;
            for (j = 0; j < max_count; j++) {
                const int inx = cv_floor_f(nextx), iny = cv_floor_f(nexty);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) { if (level == 0) status[p] = 0; break; }
                a = nextx - inx; b = nexty - iny;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float fb1 = 0, fb2 = 0; int64_t qb1 = 0, qb2 = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int gx = inx + x, gy = iny + y;
                        const int diff = CV_DESCALE(lk_pix(J, gx, gy) * iw00 + lk_pix(J, gx + 1, gy) * iw01 + lk_pix(J, gx, gy + 1) * iw10 + lk_pix(J, gx + 1, gy + 1) * iw11, W_BITS1 - 5)
                                         - Iwin[y * win + x];
                        fb1 += (float)(diff * dIwin[(y * win + x) * 2]); fb2 += (float)(diff * dIwin[(y * win + x) * 2 + 1]);
                        qb1 += (int64_t)diff * dIwin[(y * win + x) * 2]; qb2 += (int64_t)diff * dIwin[(y * win + x) * 2 + 1];
                    }
                float b1, b2;
                if (acc_mode == 0) { b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE; }
                else { b1 = (float)qb1 * FLT_SCALE; b2 = (float)qb2 * FLT_SCALE; }
                const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                nextx += dx; nexty += dy;
                next_pts[2 * p] = nextx + half; next_pts[2 * p + 1] = nexty + half;
                if (iters) iters[(size_t)p * nl + level] = j + 1;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                    next_pts[2 * p] -= dx * 0.5f; next_pts[2 * p + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[p] && level == 0) {                      /* the err block's bounds test (it can clear status; err itself is not needed) */
                const float qx = next_pts[2 * p] - half, qy = next_pts[2 * p + 1] - half;
                const int ix = cv_floor_f(qx), iy = cv_floor_f(qy);
                if (ix < -win || ix >= J->w || iy < -win || iy >= J->h) status[p] = 0;
            }
        }
    }
    free(Iwin);
    lk_free(LI, nl); lk_free(LJ, nl);
    return nl;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Frame.cc:454-472: which pairs go into findFundamentalMat.  boxes = vPreFramePotentialDynamicBorder (x, y, w, h);
 * isInDynamicRegion(prev point) uses strict inequalities (:629-640).  Returns the number of pairs written (all n pairs when the previous
 * frame had no dynamic object or at most 20 pairs survive the selection).
 * ---------------------------------------------------------------------------------------------------------------- */
int orc_fm_select(const float *cur, const float *prev, int n, int pre_have_dynamic, const float *boxes, int nboxes, float *cur_out, float *prev_out)
{
    int m = 0;
    if (pre_have_dynamic) {
        for (int i = 0; i < n; i++) {
            const float x = prev[2 * i], y = prev[2 * i + 1];
            int in = 0;
            for (int k = 0; k < nboxes; k++) {
                const float *r = boxes + 4 * k;
                if (x > r[0] && x < r[0] + r[2] && y > r[1] && y < r[1] + r[3]) { in = 1; break; }
            }
            if (!in) { cur_out[2 * m] = cur[2 * i]; cur_out[2 * m + 1] = cur[2 * i + 1]; prev_out[2 * m] = x; prev_out[2 * m + 1] = y; m++; }
        }
        if (m > 20) return m;
    }
    memcpy(cur_out, cur, sizeof(float) * 2 * (size_t)n); memcpy(prev_out, prev, sizeof(float) * 2 * (size_t)n);
    return n;
}

/* ------------------------------------------------------------------------------------------------------------------
 * cv::findFundamentalMat(points1, points2, FM_RANSAC, 1.0, 0.99)  — calib3d/src/fundam.cpp, ptsetreg.cpp
 * ---------------------------------------------------------------------------------------------------------------- */
/* cv::RNG (multiply-with-carry), RNG::uniform(int a, int b) */
typedef struct { uint64_t state; } orc_rng;
static inline unsigned orc_rng_next(orc_rng *r) { r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32); return (unsigned)r->state; }
static inline int orc_rng_uniform(orc_rng *r, int a, int b) { return a == b ? a : (int)(orc_rng_next(r) % (unsigned)(b - a) + a); }

/* cv::solveCubic (core/src/mathfuncs.cpp), coefficients c[0] x^3 + c[1] x^2 + c[2] x + c[3] */
int orc_solve_cubic(const double *c, double *roots)
{
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    double x0 = 0., x1 = 0., x2 = 0.;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x0 = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
        double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            double theta = acos(R / sqrt(Qcubed));
            double sqrtQ = sqrt(Q);
            double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * 3.1415926535897932384626433832795 / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * 3.1415926535897932384626433832795 / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) { x0 = -2 * pow(R, 1. / 3) - a1 / 3; x1 = pow(R, 1. / 3) - a1 / 3; }
            else { x0 = 2 * pow(-R, 1. / 3) - a1 / 3; x1 = -pow(-R, 1. / 3) - a1 / 3; }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            double e;
            d = sqrt(-d);
            e = pow(d + fabs(R), 1. / 3);
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    roots[0] = x0; roots[1] = x1; roots[2] = x2;
    return n;
}

/* Orthonormal basis (f1, f2) of the null space of the 7x9 system.  OpenCV takes the last two rows of Vt from its Jacobi SVD
 * (SVDecomp(A, W, U, Vt, MODIFY_A | FULL_UV)) — which pair of orthonormal vectors that yields is an implementation detail of
 * cv::SVD and cannot be pinned here; the SET of fundamental matrices run7Point returns does not depend on the basis (the pencil
 * lambda*f1 + (1-lambda)*f2 is the same plane, every root is normalised to F[8] = 1).  Here: Householder QR of A^T (9x7), the last
 * two columns of Q. */
static void null_space_7x9(const double *a /* 7x9 row-major */, double *f1, double *f2)
{
    double M[9 * 7], beta[7];
    for (int i = 0; i < 7; i++) for (int j = 0; j < 9; j++) M[j * 7 + i] = a[i * 9 + j];
    for (int k = 0; k < 7; k++) {
        double nrm = 0; for (int i = k; i < 9; i++) nrm += M[i * 7 + k] * M[i * 7 + k];
        nrm = sqrt(nrm);
        beta[k] = 0;
        if (nrm == 0) continue;
        M[k * 7 + k] += M[k * 7 + k] >= 0 ? nrm : -nrm;                 /* column k, rows k..8 now hold the Householder vector v_k */
        double vv = 0; for (int i = k; i < 9; i++) vv += M[i * 7 + k] * M[i * 7 + k];
        if (vv == 0) continue;
        beta[k] = 2. / vv;
        for (int j = k + 1; j < 7; j++) {
            double s = 0; for (int i = k; i < 9; i++) s += M[i * 7 + k] * M[i * 7 + j];
            s *= beta[k];
            for (int i = k; i < 9; i++) M[i * 7 + j] -= s * M[i * 7 + k];
        }
    }
    for (int i = 0; i < 9; i++) { f1[i] = i == 7; f2[i] = i == 8; }     /* q_j = H_0 H_1 ... H_6 e_j */
    for (int k = 6; k >= 0; k--) {
        double s1 = 0, s2 = 0;
        for (int i = k; i < 9; i++) { s1 += M[i * 7 + k] * f1[i]; s2 += M[i * 7 + k] * f2[i]; }
        s1 *= beta[k]; s2 *= beta[k];
        for (int i = k; i < 9; i++) { f1[i] -= s1 * M[i * 7 + k]; f2[i] -= s2 * M[i * 7 + k]; }
    }
}

/* run7Point (fundam.cpp): m1, m2 = 7 point pairs; returns the number of solutions (1..3) written to fmatrix (n x 9) */
int orc_fm_run7point(const float *m1, const float *m2, double *fmatrix)
{
    double a[7 * 9], c[4], r[3] = { 0, 0, 0 }, f1[9], f2[9];
    for (int i = 0; i < 7; i++) {
        const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
        a[i * 9 + 0] = x1 * x0; a[i * 9 + 1] = x1 * y0; a[i * 9 + 2] = x1;
        a[i * 9 + 3] = y1 * x0; a[i * 9 + 4] = y1 * y0; a[i * 9 + 5] = y1;
        a[i * 9 + 6] = x0; a[i * 9 + 7] = y0; a[i * 9 + 8] = 1;
    }
    null_space_7x9(a, f1, f2);
    for (int i = 0; i < 9; i++) f1[i] -= f2[i];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 -
           f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) + f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7]; t1 = f1[3] * f1[8] - f1[5] * f1[6]; t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 -
           f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) + f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    const int n = orc_solve_cubic(c, r);
    if (n < 1 || n > 3) return n;
    for (int k = 0; k < n; k++, fmatrix += 9) {
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; fmatrix[8] = 1.; }
        else fmatrix[8] = 0.;
        for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

/* FMEstimatorCallback::computeError: squared symmetric epipolar distance, stored as float */
static inline float fm_error(const double *F, float m1x, float m1y, float m2x, float m2y)
{
    double a, b, c, d1, d2, s1, s2;
    a = F[0] * m1x + F[1] * m1y + F[2];
    b = F[3] * m1x + F[4] * m1y + F[5];
    c = F[6] * m1x + F[7] * m1y + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = m2x * a + m2y * b + c;
    a = F[0] * m2x + F[3] * m2y + F[6];
    b = F[1] * m2x + F[4] * m2y + F[7];
    c = F[2] * m2x + F[5] * m2y + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = m1x * a + m1y * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 < e2 ? e2 : e1);       /* std::max(d1*d1*s1, d2*d2*s2) = (a < b) ? b : a */
}

/* haveCollinearPoints(m, count) (ptsetreg.cpp) */
static int have_collinear(const float *p, int count)
{
    const int i = count - 1;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return 1;
        }
    }
    return 0;
}

/* RANSACPointSetRegistrator::getSubset, modelPoints = 7, maxAttempts = 10000.  The 3.x constructor sets checkPartialSubsets = false:
 * seven distinct indices are drawn, then checkSubset(ms1, ms2, 7) — which only tests whether the LAST point lies on a line through two
 * earlier ones (haveCollinearPoints looks at i = count-1) — accepts the subset or the whole draw is repeated.  `partial` != 0 restates the
 * other branch of the same function (check after every added point, random back-off), kept for completeness. */
static int fm_get_subset(const float *m1, const float *m2, int count, orc_rng *rng, float *ms1, float *ms2, int *idx_out, int partial)
{
    int idx[7], i = 0, j, iters = 0;
    const int max_attempts = 10000;
    for (; iters < max_attempts; iters++) {
        for (i = 0; i < 7 && iters < max_attempts;) {
            int idx_i = 0;
            for (;;) {
                idx_i = idx[i] = orc_rng_uniform(rng, 0, count);
                for (j = 0; j < i; j++) if (idx_i == idx[j]) break;
                if (j == i) break;
            }
            ms1[2 * i] = m1[2 * idx_i]; ms1[2 * i + 1] = m1[2 * idx_i + 1];
            ms2[2 * i] = m2[2 * idx_i]; ms2[2 * i + 1] = m2[2 * idx_i + 1];
            if (partial && (have_collinear(ms1, i + 1) || have_collinear(ms2, i + 1))) {      /* !cb->checkSubset(ms1, ms2, i+1) */
                i = orc_rng_uniform(rng, 0, i + 1);
                iters++;
                continue;
            }
            i++;
        }
        if (!partial && i == 7 && (have_collinear(ms1, i) || have_collinear(ms2, i))) continue;
        break;
    }
    if (idx_out) memcpy(idx_out, idx, sizeof idx);
    return i == 7 && iters < max_attempts;
}

/* RANSACUpdateNumIters (ptsetreg.cpp) */
int orc_ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : cv_round_d(num / denom);
}

static int g_partial_subsets = 0;
void orc_fm_set_partial_subsets(int on) { g_partial_subsets = on; }
/* sample sequence tap: the 7 indices of the first `iters` RANSAC draws */
int orc_fm_sample_sequence(const float *m1, const float *m2, int n, int iters, int32_t *idx_out)
{
    orc_rng rng; rng.state = (uint64_t)-1; float a[14], b[14]; int k;
    for (k = 0; k < iters; k++) { int id[7]; if (!fm_get_subset(m1, m2, n, &rng, a, b, id, g_partial_subsets)) break; for (int j = 0; j < 7; j++) idx_out[7 * k + j] = id[j]; }
    return k;
}
/* LMeDSPointSetRegistrator::run (OpenCV ptsetreg.cpp) for the fundamental-matrix callback, modelPoints = 7, maxIters = 1000: niters = max(RANSACUpdateNumIters(confidence,
 * 0.45, 7, 1000), 3) subsets from cv::RNG(-1); per model the median = element count/2 of the sorted float errors (std::nth_element over their bit patterns, equal to float
 * order for the non-negative errors); the smallest median wins (strict '<', first wins); sigma = max(2.5 * 1.4826 * (1 + 5 / (count - 7)) * sqrt(minMedian), 0.001),
 * inliers = err <= (float)(sigma^2); the call succeeds when at least 7 inliers remain.  stats: iterations run, winning iteration, winning root, inliers. */
static int fm_lmeds(const float *m1, const float *m2, int n, double confidence, double *F, uint8_t *mask, int32_t *stats)
{
    orc_rng rng; rng.state = (uint64_t)-1;
    int niters = orc_ransac_update_num_iters(confidence, 0.45, 7, 1000); if (niters < 3) niters = 3;
    double minMedian = DBL_MAX, best[9]; memset(best, 0, sizeof best);
    float ms1[14], ms2[14], err[16]; int iter;
    for (iter = 0; iter < niters; iter++) {
        if (!fm_get_subset(m1, m2, n, &rng, ms1, ms2, NULL, g_partial_subsets)) { if (iter == 0) return 0; break; }
        double model[27];
        const int nmodels = orc_fm_run7point(ms1, ms2, model);
        if (nmodels <= 0) continue;
        for (int i = 0; i < nmodels; i++) {
            const double *Fi = model + 9 * i;
            for (int p = 0; p < n; p++) err[p] = fm_error(Fi, m1[2 * p], m1[2 * p + 1], m2[2 * p], m2[2 * p + 1]);
            for (int a = 1; a < n; a++) { const float v = err[a]; int b = a - 1; while (b >= 0 && err[b] > v) { err[b + 1] = err[b]; b--; } err[b + 1] = v; }     /* nth_element(count/2): the sorted order has the same element there */
            const double median = (double)err[n / 2];
            if (median < minMedian) { minMedian = median; memcpy(best, Fi, sizeof best); if (stats) { stats[1] = iter; stats[2] = i; } }
        }
    }
    if (stats) stats[0] = iter;
    if (!(minMedian < DBL_MAX)) return 0;
    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * sqrt(minMedian);
    if (sigma < 0.001) sigma = 0.001;
    const float t = (float)(sigma * sigma);
    int good = 0;
    for (int p = 0; p < n; p++) { const int f = fm_error(best, m1[2 * p], m1[2 * p + 1], m2[2 * p], m2[2 * p + 1]) <= t; if (mask) mask[p] = (uint8_t)f; good += f; }
    if (stats) stats[3] = good;
    if (good < 7) return 0;
    memcpy(F, best, sizeof best);
    return 1;
}
/* findFundamentalMat(m1, m2, FM_RANSAC, threshold, confidence): returns 1 and F (3x3 row-major, F[8] = 1 or 0) or 0 (empty Mat).
 * mask (optional, n bytes) = inliers of the returned model.  stats (optional, 4 ints): iterations run, index of the winning iteration,
 * root index of the winning model, inlier count.
 * n < 7 -> empty; n == 7 -> run7Point directly (first solution is what a 3x3 read of the 9x3 result sees); 8..14 points: OpenCV switches to
 * LMedS (`(method & ~3) == FM_RANSAC && npoints >= 15` gate in fundam.cpp): fm_lmeds above. */
int orc_find_fundamental_ransac(const float *m1, const float *m2, int n, double threshold, double confidence, double *F, uint8_t *mask, int32_t *stats)
{
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
    if (n < 7) return 0;
    if (n == 7) { double f[27]; const int k = orc_fm_run7point(m1, m2, f); if (k <= 0) return 0; memcpy(F, f, sizeof(double) * 9); return 1; }
    if (threshold <= 0) threshold = 3;
    if (confidence < DBL_EPSILON || confidence > 1 - DBL_EPSILON) confidence = 0.99;
    if (n < 15) return fm_lmeds(m1, m2, n, confidence, F, mask, stats);
    const float thr2 = (float)(threshold * threshold);      /* findInliers: float t = (float)(thresh*thresh) */
    orc_rng rng; rng.state = (uint64_t)-1;
    int niters = 1000, max_good = 0, iter;
    uint8_t *cur_mask = (uint8_t *)malloc((size_t)n), *best_mask = (uint8_t *)malloc((size_t)n);
    double best[9]; memset(best, 0, sizeof best);
    float ms1[14], ms2[14];
    for (iter = 0; iter < niters; iter++) {
        if (!fm_get_subset(m1, m2, n, &rng, ms1, ms2, NULL, g_partial_subsets)) { if (iter == 0) { free(cur_mask); free(best_mask); return 0; } break; }
        double model[27];
        const int nmodels = orc_fm_run7point(ms1, ms2, model);
        if (nmodels <= 0) continue;
        for (int i = 0; i < nmodels; i++) {
            const double *Fi = model + 9 * i;
            int good = 0;
            for (int p = 0; p < n; p++) { const int f = fm_error(Fi, m1[2 * p], m1[2 * p + 1], m2[2 * p], m2[2 * p + 1]) <= thr2; cur_mask[p] = (uint8_t)f; good += f; }
            if (good > (max_good > 6 ? max_good : 6)) {
                uint8_t *t = cur_mask; cur_mask = best_mask; best_mask = t;
                memcpy(best, Fi, sizeof best);
                max_good = good;
                niters = orc_ransac_update_num_iters(confidence, (double)(n - good) / n, 7, niters);
                if (stats) { stats[1] = iter; stats[2] = i; stats[3] = good; }
            }
        }
    }
    if (stats) stats[0] = iter;
    int ok = 0;
    if (max_good > 0) { memcpy(F, best, sizeof best); if (mask) memcpy(mask, best_mask, (size_t)n); ok = 1; }
    free(cur_mask); free(best_mask);
    return ok;
}

/* test taps */
unsigned orc_rng_sequence(uint64_t seed, int count, int modulo, int32_t *out)
{
    orc_rng r; r.state = seed;
    for (int i = 0; i < count; i++) out[i] = orc_rng_uniform(&r, 0, modulo);
    return (unsigned)r.state;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Frame.cc:556-604: keep[i] = epipolar distance of (cur_i, prev_i) under F (CheckEpiLineDistToRmDynamicPoint :613-627, fp64) below 0.2 inside a
 * person box of THIS frame (isInDynamicRegion, strict) / 1.0 elsewhere.  Returns 1 when the restore rule fires (a person present and fewer than
 * 0.1 * nFeatures survivors: mvKeys is restored, :599-604) — keep[] still holds the per-point decisions.
 * ---------------------------------------------------------------------------------------------------------------- */
int orc_dynamic_mask(const float *cur, const float *prev, int n, const double *F, const float *boxes, int nboxes, int have_dynamic, int nfeatures, uint8_t *keep)
{
    int sum = 0;
    for (int i = 0; i < n; i++) {
        const float x = cur[2 * i], y = cur[2 * i + 1];
        const double a = x * F[0] + y * F[1] + F[2], b = x * F[3] + y * F[4] + F[5], c = x * F[6] + y * F[7] + F[8];
        const double dist = fabs(a * prev[2 * i] + b * prev[2 * i + 1] + c) / sqrt(a * a + b * b);
        int in = 0;
        if (have_dynamic)
            for (int k = 0; k < nboxes; k++) { const float *r = boxes + 4 * k; if (x > r[0] && x < r[0] + r[2] && y > r[1] && y < r[1] + r[3]) { in = 1; break; } }
        keep[i] = (uint8_t)(dist < (in ? 0.2 : 1.0));
        sum += keep[i];
    }
    return have_dynamic && (float)sum < (float)nfeatures * 0.1f;
}
