/* oracle/orb_oracle.c — CPU ORACLE for the ORB extractor stage.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference algorithm
 *   /root/reference/src/sg-slam/src/ORBextractor.cc   (ORBextractor::operator(), :1045-1106)
 * plus the OpenCV 3.4.15 primitives that file calls (cv::resize INTER_LINEAR, cv::FAST,
 * cv::GaussianBlur 7x7 sigma 2, cv::fastAtan2, cvRound).  OpenCV is NOT vendored in the
 * reference tree and is NOT installed in this image, so those primitives are restated from
 * the published OpenCV 3.4 algorithms:  ==> PARITY UNPINNED at the OpenCV boundary <==
 * (the reference ships no golden vectors for this path; see DESIGN.md "Oracle").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (sg_slam_amd/csrc) never links or calls it.
 *
 * Everything is single-threaded, scalar, integer/fp32 exactly as the reference computes it
 * (compile with -O2 -ffp-contract=off, no -ffast-math, no -march=native).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/sgx_orb_pattern.h"

#define ORC_MAX_LEVELS 16
#define PATCH_SIZE 31
#define HALF_PATCH 15
#define EDGE_TH 19

typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint; /* cv::KeyPoint, 28 B */

/* cvRound(double/float): round-half-to-even (OpenCV uses cvtsd2si / lrint). */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------------------------------
 * E1: constructor tables — ORBextractor.cc:411-471.  `scaleFactor` is stored in a double
 * member (ORBextractor.h:99) but initialised from a float argument.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor;
    float scale[ORC_MAX_LEVELS], inv_scale[ORC_MAX_LEVELS], sigma2[ORC_MAX_LEVELS], inv_sigma2[ORC_MAX_LEVELS];
    int per_level[ORC_MAX_LEVELS];
    int umax[HALF_PATCH + 2];
} orc_params;

void orc_orb_params(orc_params *p, int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh)
{
    memset(p, 0, sizeof *p);
    p->nfeatures = nfeatures; p->nlevels = nlevels; p->iniTh = iniTh; p->minTh = minTh;
    p->scaleFactor = (double)scaleFactor;
    p->scale[0] = 1.0f; p->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {                       /* :420-424 */
        p->scale[i] = (float)((double)p->scale[i - 1] * p->scaleFactor);
        p->sigma2[i] = p->scale[i] * p->scale[i];
    }
    for (int i = 0; i < nlevels; i++) {                       /* :428-432 */
        p->inv_scale[i] = 1.0f / p->scale[i];
        p->inv_sigma2[i] = 1.0f / p->sigma2[i];
    }
    float factor = (float)(1.0f / p->scaleFactor);            /* :437 (float/double -> double -> float) */
    float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels)); /* :438 */
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {                   /* :441-446 */
        p->per_level[l] = cv_round_f(desired);
        sum += p->per_level[l];
        desired *= factor;
    }
    p->per_level[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;   /* :447 */

    /* umax: :455-470 */
    int vmax = (int)floor(HALF_PATCH * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (int v = 0; v <= vmax; ++v) p->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (int v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (p->umax[v0] == p->umax[v0 + 1]) ++v0;
        p->umax[v] = v0;
        ++v0;
    }
}

/* level size: ORBextractor.cc:1112-1113 */
void orc_level_size(const orc_params *p, int w, int h, int level, int *lw, int *lh)
{
    float s = p->inv_scale[level];
    *lw = cv_round_f((float)w * s);
    *lh = cv_round_f((float)h * s);
}

/* ------------------------------------------------------------------------------------------
 * cv::resize(u8, INTER_LINEAR) — OpenCV 3.4 imgproc/resize.cpp, generic fixed-point path
 * (HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<22>>).
 * Called at ORBextractor.cc:1121 (level l from level l-1).  SURVEY Appendix A.2.
 * ---------------------------------------------------------------------------------------- */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride)
{
    double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *alpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *rows[2];
    rows[0] = (int *)malloc(sizeof(int) * dw);
    rows[1] = (int *)malloc(sizeof(int) * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        int a0 = cv_round_f(c0 * 2048), a1 = cv_round_f(c1 * 2048);
        alpha[2 * dx] = (short)a0; alpha[2 * dx + 1] = (short)a1;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        short b0 = (short)cv_round_f((1.f - fy) * 2048), b1 = (short)cv_round_f(fy * 2048);
        for (int k = 0; k < 2; k++) {
            int r = sy + k; if (r < 0) r = 0; if (r > sh - 1) r = sh - 1;
            const uint8_t *S = src + (size_t)r * sstride;
            int *D = rows[k];
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx];
                int s1 = sx + 1 < sw ? S[sx + 1] : 0;           /* weight is 0 there */
                D[dx] = S[sx] * alpha[2 * dx] + s1 * alpha[2 * dx + 1];
            }
        }
        uint8_t *o = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            o[dx] = (uint8_t)((((b0 * (rows[0][dx] >> 4)) >> 16) + ((b1 * (rows[1][dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(alpha); free(rows[0]); free(rows[1]);
}

/* BORDER_REFLECT_101 index */
static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}

/* ------------------------------------------------------------------------------------------
 * cv::GaussianBlur(u8, 7x7, sigma 2, BORDER_REFLECT_101) — OpenCV 3.4.15 fixed-point path
 * (smooth.cpp fixedSmoothInvoker<uint8_t, ufixedpoint16>): 8.8 taps from the error-diffused
 * kernel {18,34,48,56,48,34,18}/256, horizontal pass -> 8.8, vertical -> 16.16, round.
 * Called at ORBextractor.cc:1086-1087 on a clone of the level (non-submatrix).  Appendix A.3.
 * ---------------------------------------------------------------------------------------- */
static const int GK[7] = {18, 34, 48, 56, 48, 34, 18};

void orc_gaussian7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            unsigned acc = 0;
            for (int k = -3; k <= 3; k++) acc += GK[k + 3] * src[(size_t)y * sstride + reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = (uint16_t)acc;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int k = -3; k <= 3; k++) acc += (uint32_t)GK[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((acc + 32768u) >> 16);
        }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------
 * cv::FAST(img, kps, threshold, nonmax=true) == FAST-9/16 — OpenCV 3.4 features2d/fast.cpp
 * FAST_t<16> + fast_score.cpp cornerScore<16>.  Called per cell at ORBextractor.cc:810,815.
 * Output in row-major order; response = score.  SURVEY Appendix A.1.
 * ---------------------------------------------------------------------------------------- */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int fast_is_corner(const uint8_t *p, const int *off, int t)
{
    int v = p[0];
    int dark = 0, bright = 0;  /* run lengths over the 25-entry circular walk */
    for (int k = 0; k < 25; k++) {
        int x = p[off[k & 15]];
        if (x < v - t) { if (++dark > 8) return 1; } else dark = 0;
        if (x > v + t) { if (++bright > 8) return 1; } else bright = 0;
    }
    return 0;
}

static int fast_corner_score(const uint8_t *p, const int *off, int threshold)
{
    int d[25], v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[off[k & 15]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1]; if (d[k + 2] < a) a = d[k + 2]; if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) if (d[k + j] < a) a = d[k + j];
        int m = a < d[k] ? a : d[k];       if (m > a0) a0 = m;
        m = a < d[k + 9] ? a : d[k + 9];   if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1];
        for (int j = 2; j <= 5; j++) if (d[k + j] > b) b = d[k + j];
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) if (d[k + j] > b) b = d[k + j];
        int m = b > d[k] ? b : d[k];       if (m < b0) b0 = m;
        m = b > d[k + 9] ? b : d[k + 9];   if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

/* returns number of keypoints written (x,y,score); img is cols x rows with row stride. */
int orc_fast9_16(const uint8_t *img, int stride, int cols, int rows, int threshold, int nonmax,
                 int *ox, int *oy, int *oscore, int cap)
{
    int off[16], n = 0;
    for (int k = 0; k < 16; k++) off[k] = RING_DX[k] + RING_DY[k] * stride;
    if (threshold < 0) threshold = 0; if (threshold > 255) threshold = 255;
    if (cols < 7 || rows < 7) return 0;
    uint8_t *score = (uint8_t *)calloc((size_t)cols * rows, 1);
    uint8_t *isc = (uint8_t *)calloc((size_t)cols * rows, 1);
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            const uint8_t *p = img + (size_t)y * stride + x;
            if (fast_is_corner(p, off, threshold)) {
                isc[(size_t)y * cols + x] = 1;
                score[(size_t)y * cols + x] = (uint8_t)fast_corner_score(p, off, threshold);
            }
        }
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            size_t i = (size_t)y * cols + x;
            if (!isc[i]) continue;
            int s = score[i];
            if (nonmax) {
                if (!(s > score[i - 1] && s > score[i + 1] &&
                      s > score[i - cols - 1] && s > score[i - cols] && s > score[i - cols + 1] &&
                      s > score[i + cols - 1] && s > score[i + cols] && s > score[i + cols + 1])) continue;
            }
            if (n < cap) { ox[n] = x; oy[n] = y; oscore[n] = s; }
            n++;
        }
    free(score); free(isc);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * E4 (first half): per-cell FAST over one level — ORBextractor.cc:772-830.
 * Output: candidates in cell-raster order, coordinates relative to (minBorderX,minBorderY)
 * exactly as pushed into vToDistributeKeys (:823-825).
 * ---------------------------------------------------------------------------------------- */
int orc_level_candidates(const uint8_t *img, int stride, int cols, int rows, int iniTh, int minTh,
                         float *cx, float *cy, float *cresp, int cap)
{
    const float W = 30;
    const int minBX = EDGE_TH - 3, minBY = minBX;
    const int maxBX = cols - EDGE_TH + 3, maxBY = rows - EDGE_TH + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) return 0;
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int n = 0;
    int tcap = (wCell + 6) * (hCell + 6);
    int *tx = (int *)malloc(sizeof(int) * tcap), *ty = (int *)malloc(sizeof(int) * tcap), *ts = (int *)malloc(sizeof(int) * tcap);
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
            const uint8_t *sub = img + (size_t)y0 * stride + x0;
            int m = orc_fast9_16(sub, stride, cw, ch, iniTh, 1, tx, ty, ts, tcap);
            if (m == 0) m = orc_fast9_16(sub, stride, cw, ch, minTh, 1, tx, ty, ts, tcap);
            for (int k = 0; k < m; k++) {
                if (n < cap) {
                    cx[n] = (float)tx[k] + (float)(j * wCell);
                    cy[n] = (float)ty[k] + (float)(i * hCell);
                    cresp[n] = (float)ts[k];
                }
                n++;
            }
        }
    }
    free(tx); free(ty); free(ts);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * E5: DistributeOctTree — ORBextractor.cc:540-764, DivideNode :482-538.
 * Nodes live in a pool and are chained in a doubly linked list that mirrors std::list
 * (push_front / erase).  The reference sorts (size, ExtractorNode*) pairs (:685); on equal
 * sizes the order is the order of heap addresses, which is allocator-state dependent in the
 * reference itself.  The oracle (and the product) define the tie-break as creation order:
 * a node created later compares greater (a monotonically growing heap).
 * Output: indices into the candidate arrays, in the reference's output (list) order.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int *keys; int nkeys;
    int no_more;
    int prev, next;
    int seq;
} onode;

typedef struct { onode *pool; int npool, cappool; int head, tail, size; int seq; } olist;

static int ol_new(olist *L)
{
    if (L->npool == L->cappool) { L->cappool = L->cappool ? L->cappool * 2 : 256; L->pool = (onode *)realloc(L->pool, sizeof(onode) * L->cappool); }
    onode *n = &L->pool[L->npool]; memset(n, 0, sizeof *n); n->prev = n->next = -1; n->seq = L->seq++;
    return L->npool++;
}
static void ol_push_front(olist *L, int id) { onode *n = &L->pool[id]; n->prev = -1; n->next = L->head; if (L->head >= 0) L->pool[L->head].prev = id; else L->tail = id; L->head = id; L->size++; }
static void ol_push_back(olist *L, int id) { onode *n = &L->pool[id]; n->next = -1; n->prev = L->tail; if (L->tail >= 0) L->pool[L->tail].next = id; else L->head = id; L->tail = id; L->size++; }
static int ol_erase(olist *L, int id)
{
    onode *n = &L->pool[id]; int nx = n->next;
    if (n->prev >= 0) L->pool[n->prev].next = n->next; else L->head = n->next;
    if (n->next >= 0) L->pool[n->next].prev = n->prev; else L->tail = n->prev;
    L->size--; free(n->keys); n->keys = NULL; n->nkeys = 0;
    return nx;
}

/* split node `id` into four children; returns child ids (or -1 when a child is empty and was not created) */
static void divide_node(olist *L, int id, const float *kx, const float *ky, int child[4])
{
    onode P = L->pool[id];
    const int halfX = (int)ceilf((float)(P.URx - P.ULx) / 2);
    const int halfY = (int)ceilf((float)(P.BRy - P.ULy) / 2);
    int bx[4][8];
    /* n1 */ bx[0][0] = P.ULx; bx[0][1] = P.ULy; bx[0][2] = P.ULx + halfX; bx[0][3] = P.ULy; bx[0][4] = P.ULx; bx[0][5] = P.ULy + halfY; bx[0][6] = P.ULx + halfX; bx[0][7] = P.ULy + halfY;
    /* n2 */ bx[1][0] = bx[0][2]; bx[1][1] = bx[0][3]; bx[1][2] = P.URx; bx[1][3] = P.URy; bx[1][4] = bx[0][6]; bx[1][5] = bx[0][7]; bx[1][6] = P.URx; bx[1][7] = P.ULy + halfY;
    /* n3 */ bx[2][0] = bx[0][4]; bx[2][1] = bx[0][5]; bx[2][2] = bx[0][6]; bx[2][3] = bx[0][7]; bx[2][4] = P.BLx; bx[2][5] = P.BLy; bx[2][6] = bx[0][6]; bx[2][7] = P.BLy;
    /* n4 */ bx[3][0] = bx[2][2]; bx[3][1] = bx[2][3]; bx[3][2] = bx[1][6]; bx[3][3] = bx[1][7]; bx[3][4] = bx[2][6]; bx[3][5] = bx[2][7]; bx[3][6] = P.BRx; bx[3][7] = P.BRy;
    int *ck[4]; int cn[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; c++) ck[c] = (int *)malloc(sizeof(int) * (P.nkeys > 0 ? P.nkeys : 1));
    const float sx = (float)bx[0][2], sy = (float)bx[0][7];
    for (int i = 0; i < P.nkeys; i++) {
        int k = P.keys[i];
        int c = (kx[k] < sx) ? ((ky[k] < sy) ? 0 : 2) : ((ky[k] < sy) ? 1 : 3);
        ck[c][cn[c]++] = k;
    }
    for (int c = 0; c < 4; c++) {
        if (cn[c] == 0) { free(ck[c]); child[c] = -1; continue; }
        int nid = ol_new(L);            /* NB: may realloc the pool; P is a copy */
        onode *n = &L->pool[nid];
        n->ULx = bx[c][0]; n->ULy = bx[c][1]; n->URx = bx[c][2]; n->URy = bx[c][3];
        n->BLx = bx[c][4]; n->BLy = bx[c][5]; n->BRx = bx[c][6]; n->BRy = bx[c][7];
        n->keys = ck[c]; n->nkeys = cn[c]; n->no_more = (cn[c] == 1);
        child[c] = nid;
    }
}

typedef struct { int size, seq, id; } osp;
static int osp_cmp(const void *a, const void *b)
{
    const osp *x = (const osp *)a, *y = (const osp *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

int orc_distribute_octree(const float *kx, const float *ky, const float *kresp, int nk,
                          int minX, int maxX, int minY, int maxY, int N, int *out_idx, int cap)
{
    olist L; memset(&L, 0, sizeof L); L.head = L.tail = -1;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) return -2;                                  /* the reference divides by nIni and indexes an empty vector here (:544-566): undefined, reported */
    const float hX = (float)(maxX - minX) / nIni;
    int *ini = (int *)malloc(sizeof(int) * (nIni > 0 ? nIni : 1));
    for (int i = 0; i < nIni; i++) {
        int id = ol_new(&L); onode *n = &L.pool[id];
        n->ULx = (int)(hX * (float)i); n->ULy = 0;
        n->URx = (int)(hX * (float)(i + 1)); n->URy = 0;
        n->BLx = n->ULx; n->BLy = maxY - minY;
        n->BRx = n->URx; n->BRy = maxY - minY;
        n->keys = (int *)malloc(sizeof(int) * (nk > 0 ? nk : 1));
        ol_push_back(&L, id); ini[i] = id;
    }
    for (int i = 0; i < nk; i++) { onode *n = &L.pool[ini[(size_t)(kx[i] / hX)]]; n->keys[n->nkeys++] = i; }
    free(ini);
    for (int it = L.head; it >= 0;) {
        onode *n = &L.pool[it];
        if (n->nkeys == 1) { n->no_more = 1; it = n->next; }
        else if (n->nkeys == 0) it = ol_erase(&L, it);
        else it = n->next;
    }
    int finish = 0;
    osp *cur = NULL, *prevv = NULL; int ncur = 0, capcur = 0, nprev = 0, capprev = 0;
#define PUSH_CUR(ID) do { if (ncur == capcur) { capcur = capcur ? capcur * 2 : 256; cur = (osp *)realloc(cur, sizeof(osp) * capcur); } \
        cur[ncur].size = L.pool[ID].nkeys; cur[ncur].seq = L.pool[ID].seq; cur[ncur].id = (ID); ncur++; } while (0)
    while (!finish) {
        int prevSize = L.size, nToExpand = 0;
        ncur = 0;
        for (int it = L.head; it >= 0;) {
            if (L.pool[it].no_more) { it = L.pool[it].next; continue; }
            int ch[4]; divide_node(&L, it, kx, ky, ch);
            for (int c = 0; c < 4; c++) if (ch[c] >= 0) {
                ol_push_front(&L, ch[c]);
                if (L.pool[ch[c]].nkeys > 1) { nToExpand++; PUSH_CUR(ch[c]); }
            }
            it = ol_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) finish = 1;
        else if (L.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = L.size;
                if (capprev < ncur) { capprev = ncur; prevv = (osp *)realloc(prevv, sizeof(osp) * (capprev ? capprev : 1)); }
                memcpy(prevv, cur, sizeof(osp) * ncur); nprev = ncur; ncur = 0;
                qsort(prevv, nprev, sizeof(osp), osp_cmp);   /* total order (seq unique) == stable_sort on pairs */
                for (int j = nprev - 1; j >= 0; j--) {
                    int ch[4]; divide_node(&L, prevv[j].id, kx, ky, ch);
                    for (int c = 0; c < 4; c++) if (ch[c] >= 0) {
                        ol_push_front(&L, ch[c]);
                        if (L.pool[ch[c]].nkeys > 1) PUSH_CUR(ch[c]);
                    }
                    ol_erase(&L, prevv[j].id);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
#undef PUSH_CUR
    int n = 0;
    for (int it = L.head; it >= 0; it = L.pool[it].next) {     /* :745-761 */
        onode *nd = &L.pool[it];
        int best = nd->keys[0]; float mr = kresp[best];
        for (int k = 1; k < nd->nkeys; k++) if (kresp[nd->keys[k]] > mr) { best = nd->keys[k]; mr = kresp[best]; }
        if (n < cap) out_idx[n] = best;
        n++;
    }
    for (int i = 0; i < L.npool; i++) free(L.pool[i].keys);
    free(L.pool); free(cur); free(prevv);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * cv::fastAtan2 (degrees) — OpenCV 3.4 core/mathfuncs_core (atan_f32), Appendix A.4.
 * ---------------------------------------------------------------------------------------- */
float orc_fast_atan2(float y, float x)
{
    const float k = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-16;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay / (ax + eps); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + eps); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* E6: IC_Angle — ORBextractor.cc:78-105 */
float orc_ic_angle(const uint8_t *img, int stride, float px, float py, const int *umax)
{
    int m01 = 0, m10 = 0;
    const uint8_t *c = img + (size_t)cv_round_f(py) * stride + cv_round_f(px);
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m10 += u * c[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int vsum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int vp = c[u + v * stride], vm = c[u - v * stride];
            vsum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    return orc_fast_atan2((float)m01, (float)m10);
}

/* E8: computeOrbDescriptor — ORBextractor.cc:108-148.  cos/sin are the float overloads
 * (std::cos(float) via `using namespace std`), i.e. the host libm cosf/sinf. */
void orc_descriptor(const uint8_t *blur, int stride, float px, float py, float angle_deg, uint8_t *desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t *c = blur + (size_t)cv_round_f(py) * stride + cv_round_f(px);
    const signed char *pt = sgx_orb_pattern_xy;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++, pt += 4) {
            float x0 = pt[0], y0 = pt[1], x1 = pt[2], y1 = pt[3];
            int t0 = c[cv_round_f(x0 * b + y0 * a) * stride + cv_round_f(x0 * a - y0 * b)];
            int t1 = c[cv_round_f(x1 * b + y1 * a) * stride + cv_round_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------------------------
 * E2: ORBextractor::operator() — ORBextractor.cc:1045-1106 (+ ComputePyramid :1108-1133,
 * ComputeKeyPointsOctTree :766-854).  Borders (copyMakeBorder) are not materialised: with
 * EDGE_THRESHOLD 19 no consumer on the RGB-D path reads outside a level's own pixels except
 * the blur, which applies REFLECT_101 itself on the cloned level (:1086-1087).
 * Optional debug outputs: pyr (concatenated levels, row stride = level width) and per-level
 * candidate counts.
 * ---------------------------------------------------------------------------------------- */
int orc_orb_extract(const uint8_t *gray, int w, int h, int stride,
                    int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh,
                    orc_keypoint *kps, uint8_t *desc, int cap,
                    uint8_t *pyr_out, int *ncand_out)
{
    orc_params P; orc_orb_params(&P, nfeatures, scaleFactor, nlevels, iniTh, minTh);
    uint8_t *lv[ORC_MAX_LEVELS]; int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    size_t pyr_off = 0;
    for (int l = 0; l < nlevels; l++) {           /* geometries on which the reference's DistributeOctTree is undefined (nIni = 0, :544-566): reported, like the product's create */
        int lw_, lh_; orc_level_size(&P, w, h, l, &lw_, &lh_);
        const int bw = lw_ - 2 * (EDGE_TH - 3), bh = lh_ - 2 * (EDGE_TH - 3);
        if (bw < 1 || bh < 1 || (int)roundf((float)bw / (float)bh) < 1) return -2;
    }
    for (int l = 0; l < nlevels; l++) {
        orc_level_size(&P, w, h, l, &lw[l], &lh[l]);
        lv[l] = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
        if (l == 0) for (int y = 0; y < h; y++) memcpy(lv[0] + (size_t)y * w, gray + (size_t)y * stride, w);
        else orc_resize_linear_u8(lv[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lv[l], lw[l], lh[l], lw[l]);
        if (pyr_out) { memcpy(pyr_out + pyr_off, lv[l], (size_t)lw[l] * lh[l]); pyr_off += (size_t)lw[l] * lh[l]; }
    }
    int total = 0;
    int ccap = w * h / 4 + 16;
    float *cx = (float *)malloc(sizeof(float) * ccap), *cy = (float *)malloc(sizeof(float) * ccap), *cr = (float *)malloc(sizeof(float) * ccap);
    int *sel = (int *)malloc(sizeof(int) * (nfeatures * 4 + 16));
    for (int l = 0; l < nlevels; l++) {
        int nc = orc_level_candidates(lv[l], lw[l], lw[l], lh[l], iniTh, minTh, cx, cy, cr, ccap);
        if (ncand_out) ncand_out[l] = nc;
        if (nc > ccap) nc = ccap;
        const int minBX = EDGE_TH - 3, minBY = minBX, maxBX = lw[l] - EDGE_TH + 3, maxBY = lh[l] - EDGE_TH + 3;
        int ns = nc > 0 ? orc_distribute_octree(cx, cy, cr, nc, minBX, maxBX, minBY, maxBY, P.per_level[l], sel, nfeatures * 4 + 16) : 0;
        if (ns == 0) continue;
        uint8_t *blur = (uint8_t *)malloc((size_t)lw[l] * lh[l]);
        orc_gaussian7_u8(lv[l], lw[l], lh[l], lw[l], blur, lw[l]);
        const int scaledPatch = (int)(PATCH_SIZE * P.scale[l]);           /* :838 float -> int */
        for (int i = 0; i < ns; i++) {
            int k = sel[i];
            float x = cx[k] + minBX, y = cy[k] + minBY;                    /* :844-845 */
            float ang = orc_ic_angle(lv[l], lw[l], x, y, P.umax);
            if (total < cap) {
                orc_keypoint *kp = &kps[total];
                orc_descriptor(blur, lw[l], x, y, ang, desc + (size_t)total * 32);
                kp->x = x; kp->y = y;
                if (l != 0) { kp->x = x * P.scale[l]; kp->y = y * P.scale[l]; }   /* :1096-1102 */
                kp->size = (float)scaledPatch; kp->angle = ang; kp->response = cr[k];
                kp->octave = l; kp->class_id = -1;
            }
            total++;
        }
        free(blur);
    }
    for (int l = 0; l < nlevels; l++) free(lv[l]);
    free(cx); free(cy); free(cr); free(sel);
    return total;
}

/* flat accessor for tests (ctypes): fills arrays from orc_orb_params */
void orc_orb_params_flat(int nfeatures, float scaleFactor, int nlevels, float *scale, float *inv_scale,
                         float *sigma2, float *inv_sigma2, int *per_level, int *umax16)
{
    orc_params P; orc_orb_params(&P, nfeatures, scaleFactor, nlevels, 20, 7);
    for (int i = 0; i < nlevels; i++) { scale[i] = P.scale[i]; inv_scale[i] = P.inv_scale[i]; sigma2[i] = P.sigma2[i]; inv_sigma2[i] = P.inv_sigma2[i]; per_level[i] = P.per_level[i]; }
    for (int i = 0; i <= HALF_PATCH; i++) umax16[i] = P.umax[i];
}
