/* oracle/match_oracle.c — CPU ORACLE for the ORB matcher stage.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates, with the reference's own sequential/greedy semantics:
 *   ORBmatcher::DescriptorDistance            src/sg-slam/src/ORBmatcher.cc:1649-1665
 *   ORBmatcher::SearchByProjection(cur,last)  src/sg-slam/src/ORBmatcher.cc:1332-1472
 *   ORBmatcher::ComputeThreeMaxima            src/sg-slam/src/ORBmatcher.cc:1603-1644
 *   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea   src/sg-slam/src/Frame.cc:257-272,409-419,354-407
 *   Frame::ComputeStereoFromRGBD / UnprojectStereo                src/sg-slam/src/Frame.cc:893-932
 * The 3x3 float cv::Mat products the reference uses (Rcw*x3Dw+tcw etc.) go through OpenCV's
 * cv::gemm small-matrix path (not in the reference tree): t = a0*b0+a1*b1+a2*b2 in float, then
 * (float)(t*alpha + beta*c) in double.   ==> PARITY UNPINNED at that boundary <==
 * Only tests/, smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <float.h>

typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint;

#define GRID_COLS 64
#define GRID_ROWS 48
#define TH_HIGH 100
#define HISTO_LENGTH 30
#define TH_LOW 50                  /* ORBmatcher::TH_LOW, ORBmatcher.cc:38 */

int orc_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    const uint32_t *pa = (const uint32_t *)a, *pb = (const uint32_t *)b;
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* cv::gemm small path: d = (float)((a_row . b)*alpha + beta*c), dot in float left-to-right */
static float gemm3(const float *arow, const float *b, double alpha, double beta, float c)
{
    float t = arow[0] * b[0] + arow[1] * b[1] + arow[2] * b[2];
    return (float)(t * alpha + beta * c);
}

typedef struct { int *idx; int n, cap; } cellvec;

typedef struct {
    cellvec cell[GRID_COLS][GRID_ROWS];
    float minX, maxX, minY, maxY, invW, invH;
} grid_t;

static void grid_build(grid_t *g, int N, const orc_keypoint *keys, float minX, float maxX, float minY, float maxY)
{
    memset(g, 0, sizeof *g);
    g->minX = minX; g->maxX = maxX; g->minY = minY; g->maxY = maxY;
    g->invW = (float)GRID_COLS / (maxX - minX);      /* Frame.cc:184-185 */
    g->invH = (float)GRID_ROWS / (maxY - minY);
    for (int i = 0; i < N; i++) {
        int px = (int)round((keys[i].x - minX) * g->invW);     /* PosInGrid: round(), Frame.cc:411-412 */
        int py = (int)round((keys[i].y - minY) * g->invH);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) continue;
        cellvec *c = &g->cell[px][py];
        if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 8; c->idx = (int *)realloc(c->idx, sizeof(int) * c->cap); }
        c->idx[c->n++] = i;
    }
}
static void grid_free(grid_t *g) { for (int i = 0; i < GRID_COLS; i++) for (int j = 0; j < GRID_ROWS; j++) free(g->cell[i][j].idx); }

/* Frame::GetFeaturesInArea, Frame.cc:354-407; returns count, indices in the reference's scan order */
static int features_in_area(const grid_t *g, const orc_keypoint *keys, float x, float y, float r, int minLevel, int maxLevel, int *out)
{
    int n = 0;
    int nMinCellX = (int)floorf((x - g->minX - r) * g->invW); if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - g->minX + r) * g->invW); if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - g->minY - r) * g->invH); if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - g->minY + r) * g->invH); if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const cellvec *c = &g->cell[ix][iy];
            for (int j = 0; j < c->n; j++) {
                const orc_keypoint *kp = &keys[c->idx[j]];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float dx = kp->x - x, dy = kp->y - y;
                if (fabsf(dx) < r && fabsf(dy) < r) out[n++] = c->idx[j];
            }
        }
    return n;
}

static void three_maxima(const int *cnt, int L, int *i1, int *i2, int *i3)
{
    int m1 = 0, m2 = 0, m3 = 0; *i1 = *i2 = *i3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = cnt[i];
        if (s > m1) { m3 = m2; m2 = m1; m1 = s; *i3 = *i2; *i2 = *i1; *i1 = i; }
        else if (s > m2) { m3 = m2; m2 = s; *i3 = *i2; *i2 = i; }
        else if (s > m3) { m3 = s; *i3 = i; }
    }
    if (m2 < 0.1f * (float)m1) { *i2 = -1; *i3 = -1; }
    else if (m3 < 0.1f * (float)m1) { *i3 = -1; }
}

/* ORBmatcher::SearchByProjection(Frame &Cur, const Frame &Last, th, bMono), ORBmatcher.cc:1332-1472.
 * cur_match[k] (in/out): index of the last-frame map point held by current keypoint k, -1 = NULL.
 * l_obs[i] = MapPoint::Observations() of the last frame's i-th map point. Returns nmatches. */
int orc_search_by_projection_frame(
    int Nc, const orc_keypoint *ckeys, const uint8_t *cdesc, const float *curight, const float *cTcw,
    int Nl, const orc_keypoint *lkeys, const uint8_t *l_has_mp, const uint8_t *l_outlier, const float *l_xw,
    const int *l_obs, const uint8_t *l_mpdesc, const float *lTcw,
    float fx, float fy, float cx, float cy, float bf, float minX, float maxX, float minY, float maxY,
    const float *scale_factors, float th, int bMono, int check_ori, int *cur_match)
{
    int nmatches = 0;
    grid_t *g = (grid_t *)malloc(sizeof(grid_t));
    grid_build(g, Nc, ckeys, minX, maxX, minY, maxY);
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH], hc[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { hist[i] = NULL; hn[i] = 0; hc[i] = 0; }
    const float factor = HISTO_LENGTH / 360.0f;
    const float mb = bf / fx;                                   /* Frame.cc:196 */

    /* Rcw, tcw (row-major 4x4) */
    float Rcw[3][3], tcw[3], Rlw[3][3], tlw[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { Rcw[r][c] = cTcw[4 * r + c]; Rlw[r][c] = lTcw[4 * r + c]; } tcw[r] = cTcw[4 * r + 3]; tlw[r] = lTcw[4 * r + 3]; }
    /* twc = -Rcw.t()*tcw : gemm with a transpose flag -> generic path, double accumulation, alpha=-1 */
    float twc[3];
    for (int i = 0; i < 3; i++) {
        double s = 0; for (int k = 0; k < 3; k++) s += (double)Rcw[k][i] * (double)tcw[k];
        twc[i] = (float)(s * -1.0);
    }
    /* tlc = Rlw*twc+tlw */
    const float tlc2 = gemm3(Rlw[2], twc, 1.0, 1.0, tlw[2]);
    const int bForward = tlc2 > mb && !bMono;
    const int bBackward = -tlc2 > mb && !bMono;

    int *vind = (int *)malloc(sizeof(int) * (Nc > 0 ? Nc : 1));
    for (int i = 0; i < Nl; i++) {
        if (!l_has_mp[i] || l_outlier[i]) continue;
        const float *xw = l_xw + 3 * i;
        const float x3[3] = { gemm3(Rcw[0], xw, 1.0, 1.0, tcw[0]), gemm3(Rcw[1], xw, 1.0, 1.0, tcw[1]), gemm3(Rcw[2], xw, 1.0, 1.0, tcw[2]) };
        const float xc = x3[0], yc = x3[1];
        const float invzc = (float)(1.0 / x3[2]);
        if (invzc < 0) continue;
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (u < minX || u > maxX) continue;
        if (v < minY || v > maxY) continue;
        const int nLastOctave = lkeys[i].octave;
        const float radius = th * scale_factors[nLastOctave];
        int nv;
        if (bForward) nv = features_in_area(g, ckeys, u, v, radius, nLastOctave, -1, vind);
        else if (bBackward) nv = features_in_area(g, ckeys, u, v, radius, 0, nLastOctave, vind);
        else nv = features_in_area(g, ckeys, u, v, radius, nLastOctave - 1, nLastOctave + 1, vind);
        if (nv == 0) continue;
        const uint8_t *dMP = l_mpdesc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (int q = 0; q < nv; q++) {
            const int i2 = vind[q];
            if (cur_match[i2] >= 0 && l_obs[cur_match[i2]] > 0) continue;
            if (curight[i2] > 0) {
                const float ur = u - bf * invzc;
                const float er = fabsf(ur - curight[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(dMP, cdesc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_match[bestIdx2] = i;
            nmatches++;
            if (check_ori) {
                float rot = lkeys[i].angle - ckeys[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (hn[bin] == hc[bin]) { hc[bin] = hc[bin] ? 2 * hc[bin] : 64; hist[bin] = (int *)realloc(hist[bin], sizeof(int) * hc[bin]); }
                hist[bin][hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != i1 && i != i2 && i != i3)
                for (int j = 0; j < hn[i]; j++) { cur_match[hist[i][j]] = -1; nmatches--; }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    free(vind); grid_free(g); free(g);
    return nmatches;
}

/* Frame::ComputeStereoFromRGBD, Frame.cc:893-914.  depth is the CV_32F image after
 * imDepth.convertTo(CV_32F, mDepthMapFactor) with mDepthMapFactor = 1/DepthMapFactor (Tracking.cc:229-230, :139-142);
 * cv::Mat::at<float>(v,u) with float arguments truncates them to int. */
void orc_compute_stereo_from_rgbd(int N, const orc_keypoint *keys, const orc_keypoint *keys_un, const float *depth, int w,
                                  float bf, float *uright, float *zdepth)
{
    for (int i = 0; i < N; i++) {
        uright[i] = -1; zdepth[i] = -1;
        const float d = depth[(size_t)(int)keys[i].y * w + (int)keys[i].x];
        if (d > 0) { zdepth[i] = d; uright[i] = keys_un[i].x - bf / d; }
    }
}

/* Tracking.cc:229-230: imDepth.convertTo(imDepth, CV_32F, mDepthMapFactor): u16 -> float via
 * saturate_cast<float>(src*alpha) with alpha a double (cv::convertScale 16u->32f computes in float: src*(float)alpha). */
void orc_depth_convert(const uint16_t *raw, size_t n, float depth_map_factor_inv, float *out)
{
    for (size_t i = 0; i < n; i++) out[i] = (float)raw[i] * depth_map_factor_inv;
}

/* Frame::UnprojectStereo, Frame.cc:916-930, with mRwc = Rcw^T (cv::Mat::t() copy) and mOw = -Rcw^T*tcw (Frame.cc:288-294) */
int orc_unproject_stereo(const orc_keypoint *kp_un, float z, const float *Tcw, float fx, float fy, float cx, float cy, float *xw)
{
    if (!(z > 0)) return 0;
    const float invfx = 1.0f / fx, invfy = 1.0f / fy;
    const float x = (kp_un->x - cx) * z * invfx, y = (kp_un->y - cy) * z * invfy;
    float Rwc[3][3], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rwc[r][c] = Tcw[4 * c + r]; tcw[r] = Tcw[4 * r + 3]; }
    for (int i = 0; i < 3; i++) {              /* mOw = -mRcw.t()*mtcw : transpose flag -> generic gemm path */
        double s = 0; for (int k = 0; k < 3; k++) s += (double)Tcw[4 * k + i] * (double)tcw[k];
        Ow[i] = (float)(s * -1.0);
    }
    const float xc[3] = { x, y, z };
    for (int i = 0; i < 3; i++) xw[i] = gemm3(Rwc[i], xc, 1.0, 1.0, Ow[i]);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Local-map matcher: Frame::isInFrustum (Frame.cc:296-352), MapPoint::PredictScale (MapPoint.cc:402-417),
 * ORBmatcher::RadiusByViewingCos (ORBmatcher.cc:131-137) and
 * ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)  (ORBmatcher.cc:45-129),
 * driven as Tracking::SearchLocalPoints does (Tracking.cc:1262-1312).
 * Local map point i: xw, normal (GetNormal), min_dist / max_dist (mfMinDistance / mfMaxDistance; the invariance bounds are
 * 0.8*min and 1.2*max, MapPoint.cc:372-383), descriptor, obs, skip (isBad() or mnLastFrameSeen == current frame id).
 * cur_mp_obs[k]: Observations() of the map point keypoint k already holds (-1 = NULL) — from the motion-model stage.
 * Output: cur_match[k] = index of the local map point newly assigned to keypoint k, or -1 (unchanged); in_view[i] = mbTrackInView.
 * ---------------------------------------------------------------------------------------- */
int orc_search_by_projection_local(
    int Nc, const orc_keypoint *ckeys, const uint8_t *cdesc, const float *curight, const float *cTcw, const int *cur_mp_obs,
    int Nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc,
    const int *m_obs, const uint8_t *m_skip,
    float fx, float fy, float cx, float cy, float bf, float minX, float maxX, float minY, float maxY,
    const float *scale_factors, int nlevels, float log_scale_factor, float th, float nnratio, float viewing_cos_limit,
    int *cur_match, uint8_t *in_view)
{
    int nmatches = 0;
    grid_t *g = (grid_t *)malloc(sizeof(grid_t));
    grid_build(g, Nc, ckeys, minX, maxX, minY, maxY);
    float Rcw[3][3], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r][c] = cTcw[4 * r + c]; tcw[r] = cTcw[4 * r + 3]; }
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)Rcw[k][i] * (double)tcw[k]; Ow[i] = (float)(s * -1.0); }   /* mOw, Frame.cc:288-294 */
    int *owner_obs = (int *)malloc(sizeof(int) * (Nc > 0 ? Nc : 1));
    for (int k = 0; k < Nc; k++) { owner_obs[k] = cur_mp_obs ? cur_mp_obs[k] : -1; cur_match[k] = -1; }
    int *vind = (int *)malloc(sizeof(int) * (Nc > 0 ? Nc : 1));
    const int bFactor = th != 1.0f;
    for (int i = 0; i < Nm; i++) {
        in_view[i] = 0;
        if (m_skip[i]) continue;
        /* ---- isInFrustum */
        const float *P = m_xw + 3 * i;
        const float Pc[3] = { gemm3(Rcw[0], P, 1.0, 1.0, tcw[0]), gemm3(Rcw[1], P, 1.0, 1.0, tcw[1]), gemm3(Rcw[2], P, 1.0, 1.0, tcw[2]) };
        if (Pc[2] < 0.0f) continue;
        const float invz = 1.0f / Pc[2];
        const float u = fx * Pc[0] * invz + cx, v = fy * Pc[1] * invz + cy;
        if (u < minX || u > maxX) continue;
        if (v < minY || v > maxY) continue;
        const float maxDistance = 1.2f * m_max_dist[i], minDistance = 0.8f * m_min_dist[i];
        const float PO[3] = { P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2] };
        const float dist = (float)sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);   /* cv::norm: double accumulation */
        if (dist < minDistance || dist > maxDistance) continue;
        const float *Pn = m_normal + 3 * i;
        const float viewCos = (float)(((double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2]) / (double)dist);   /* Mat::dot returns double */
        if (viewCos < viewing_cos_limit) continue;
        const float ratio = m_max_dist[i] / dist;
        int nPredictedLevel = (int)ceilf(logf(ratio) / log_scale_factor);      /* std::log(float)/std::ceil(float): `using namespace std` reaches MapPoint.cc via TemplatedVocabulary.h:36 */
        if (nPredictedLevel < 0) nPredictedLevel = 0; else if (nPredictedLevel >= nlevels) nPredictedLevel = nlevels - 1;
        in_view[i] = 1;
        const float projXR = u - bf * invz;
        /* ---- SearchByProjection body */
        float r = viewCos > 0.998 ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        const float rad = r * scale_factors[nPredictedLevel];
        const int nv = features_in_area(g, ckeys, u, v, rad, nPredictedLevel - 1, nPredictedLevel, vind);
        if (nv == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int q = 0; q < nv; q++) {
            const int idx = vind[q];
            if (owner_obs[idx] > 0) continue;                                     /* holds a map point with observations */
            if (curight[idx] > 0) { const float er = fabsf(projXR - curight[idx]); if (er > rad) continue; }
            const int d = orc_descriptor_distance(m_desc + 32 * (size_t)i, cdesc + 32 * (size_t)idx);
            if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = ckeys[idx].octave; bestIdx = idx; }
            else if (d < bestDist2) { bestLevel2 = ckeys[idx].octave; bestDist2 = d; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            cur_match[bestIdx] = i; owner_obs[bestIdx] = m_obs[i];
            nmatches++;
        }
    }
    free(vind); free(owner_obs); grid_free(g); free(g);
    return nmatches;
}


/* ---- LocalMapping gates (tier N2) --------------------------------------------------------------------------------------------------------------- */
void orc_hamming_matrix(const uint8_t *a, int na, const uint8_t *b, int nb, uint16_t *out)
{
    for (int i = 0; i < na; i++) for (int j = 0; j < nb; j++) out[(size_t)i * nb + j] = (uint16_t)orc_descriptor_distance(a + 32 * (size_t)i, b + 32 * (size_t)j);
}

/* ORBmatcher::CheckDistEpipolarLine, ORBmatcher.cc:140-158 (F12 row-major float) */
static int check_dist_epipolar(const orc_keypoint *kp1, const orc_keypoint *kp2, const float *F12, const float *sigma2_2)
{
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2_2[kp2->octave];
}

static int cmp_int(const void *x, const void *y) { const int a = *(const int *)x, b = *(const int *)y; return (a > b) - (a < b); }

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo), ORBmatcher.cc:659-827.
 * feat_node[i] = vocabulary node (DBoW2 FeatureVector key, levelsup 4) of keypoint i, -1 = none: mFeatVec is a std::map<NodeId, vector<index>> whose vectors hold the
 * indices in ascending order (TemplatedVocabulary::transform pushes them in feature order).  cam_center1 = pKF1->GetCameraCenter(), Tcw2 = pKF2 pose.
 * pairs (out): (idx1, idx2) in ascending idx1; returns nmatches. */
int orc_search_for_triangulation(int n1, const orc_keypoint *k1, const uint8_t *d1, const float *ur1, const uint8_t *has1, const int *node1, const float *cam_center1,
                                 int n2, const orc_keypoint *k2, const uint8_t *d2, const float *ur2, const uint8_t *has2, const int *node2, const float *Tcw2,
                                 const float *F12, float fx, float fy, float cx, float cy, const float *scale2, const float *sigma2_2,
                                 int only_stereo, int check_ori, int *pairs)
{
    float C2[3];
    for (int r = 0; r < 3; r++) C2[r] = gemm3(Tcw2 + 4 * r, cam_center1, 1.0, 1.0, Tcw2[4 * r + 3]);       /* C2 = R2w*Cw + t2w (:666-670) */
    const float invz = 1.0f / C2[2];
    const float ex = fx * C2[0] * invz + cx, ey = fy * C2[1] * invz + cy;
    int nmatches = 0;
    uint8_t *matched2 = (uint8_t *)calloc(n2 > 0 ? n2 : 1, 1);
    int *m12 = (int *)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++) m12[i] = -1;
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH], hc[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { hist[i] = NULL; hn[i] = hc[i] = 0; }
    const float factor = HISTO_LENGTH / 360.0f;
    /* the ordered set of nodes both keyframes have */
    int *u1 = (int *)malloc(sizeof(int) * (n1 > 0 ? n1 : 1)), nu1 = 0;
    for (int i = 0; i < n1; i++) if (node1[i] >= 0) u1[nu1++] = node1[i];
    qsort(u1, nu1, sizeof(int), cmp_int);
    for (int q = 0; q < nu1; q++) {
        if (q > 0 && u1[q] == u1[q - 1]) continue;
        const int nid = u1[q];
        int any2 = 0; for (int j = 0; j < n2; j++) if (node2[j] == nid) { any2 = 1; break; }
        if (!any2) continue;
        for (int idx1 = 0; idx1 < n1; idx1++) {                                                           /* f1it->second in index order */
            if (node1[idx1] != nid) continue;
            if (has1[idx1]) continue;
            const int stereo1 = ur1[idx1] >= 0;
            if (only_stereo && !stereo1) continue;
            int bestDist = TH_LOW, bestIdx2 = -1;
            for (int idx2 = 0; idx2 < n2; idx2++) {
                if (node2[idx2] != nid) continue;
                if (matched2[idx2] || has2[idx2]) continue;
                const int stereo2 = ur2[idx2] >= 0;
                if (only_stereo && !stereo2) continue;
                const int dist = orc_descriptor_distance(d1 + 32 * (size_t)idx1, d2 + 32 * (size_t)idx2);
                if (dist > TH_LOW || dist > bestDist) continue;
                if (!stereo1 && !stereo2) {
                    const float distex = ex - k2[idx2].x, distey = ey - k2[idx2].y;
                    if (distex * distex + distey * distey < 100 * scale2[k2[idx2].octave]) continue;
                }
                if (check_dist_epipolar(&k1[idx1], &k2[idx2], F12, sigma2_2)) { bestIdx2 = idx2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) {
                m12[idx1] = bestIdx2; matched2[bestIdx2] = 1; nmatches++;
                if (check_ori) {
                    float rot = k1[idx1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    if (hn[bin] == hc[bin]) { hc[bin] = hc[bin] ? 2 * hc[bin] : 64; hist[bin] = (int *)realloc(hist[bin], sizeof(int) * hc[bin]); }
                    hist[bin][hn[bin]++] = idx1;
                }
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3; three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; j++) { matched2[m12[hist[i][j]]] = 0; m12[hist[i][j]] = -1; nmatches--; }
        }
    }
    int n = 0;
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) { pairs[2 * n] = i; pairs[2 * n + 1] = m12[i]; n++; }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    free(matched2); free(m12); free(u1);
    return nmatches;
}


/* ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), ORBmatcher.cc:159-290.  kf_good_mp[i] = the keyframe's keypoint i holds a map
 * point that is not bad; node_* = FeatureVector key per keypoint (-1 = none).  match_f[j] (out) = index of the keyframe keypoint whose map point keypoint j of the
 * frame receives, or -1.  Returns nmatches. */
/* shared body of the two SearchByBoW overloads: good_f (may be NULL) filters side 2, strict selects `bestDist1 < TH_LOW` (KeyFrame-KeyFrame, :597) over `<= TH_LOW` (KeyFrame-Frame, :234) */
static int search_by_bow_core(int nk, const orc_keypoint *kk, const uint8_t *dk, const uint8_t *kf_good_mp, const int *node_k,
                              int nf, const orc_keypoint *kf, const uint8_t *df, const uint8_t *good_f, const int *node_f, float nnratio, int check_ori, int strict, int *match_f)
{
    int nmatches = 0;
    for (int j = 0; j < nf; j++) match_f[j] = -1;
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH], hc[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { hist[i] = NULL; hn[i] = hc[i] = 0; }
    const float factor = HISTO_LENGTH / 360.0f;
    int *u1 = (int *)malloc(sizeof(int) * (nk > 0 ? nk : 1)), nu1 = 0;
    for (int i = 0; i < nk; i++) if (node_k[i] >= 0) u1[nu1++] = node_k[i];
    qsort(u1, nu1, sizeof(int), cmp_int);
    for (int q = 0; q < nu1; q++) {
        if (q > 0 && u1[q] == u1[q - 1]) continue;
        const int nid = u1[q];
        for (int ik = 0; ik < nk; ik++) {
            if (node_k[ik] != nid || !kf_good_mp[ik]) continue;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            for (int jf = 0; jf < nf; jf++) {
                if (node_f[jf] != nid || match_f[jf] >= 0) continue;
                if (good_f && !good_f[jf]) continue;
                const int dist = orc_descriptor_distance(dk + 32 * (size_t)ik, df + 32 * (size_t)jf);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = jf; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if ((strict ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) && (float)bestDist1 < nnratio * (float)bestDist2) {
                match_f[bestIdxF] = ik;
                if (check_ori) {
                    float rot = kk[ik].angle - kf[bestIdxF].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    if (hn[bin] == hc[bin]) { hc[bin] = hc[bin] ? 2 * hc[bin] : 64; hist[bin] = (int *)realloc(hist[bin], sizeof(int) * hc[bin]); }
                    hist[bin][hn[bin]++] = bestIdxF;
                }
                nmatches++;
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3; three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; j++) { match_f[hist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    free(u1);
    return nmatches;
}
int orc_search_by_bow(int nk, const orc_keypoint *kk, const uint8_t *dk, const uint8_t *kf_good_mp, const int *node_k,
                      int nf, const orc_keypoint *kf, const uint8_t *df, const int *node_f, float nnratio, int check_ori, int *match_f)
{
    return search_by_bow_core(nk, kk, dk, kf_good_mp, node_k, nf, kf, df, NULL, node_f, nnratio, check_ori, 0, match_f);
}
/* int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) (ORBmatcher.cc:524-655; caller LoopClosing::ComputeSim3, LoopClosing.cc:259):
 * good1 / good2 = the keypoint holds a map point that is not bad; match12[i1] (out) = keypoint of pKF2 whose map point pKF1's keypoint i1 is matched with, -1 = NULL. */
int orc_search_by_bow_kf(int n1, const orc_keypoint *k1, const uint8_t *d1, const uint8_t *good1, const int *node1,
                         int n2, const orc_keypoint *k2, const uint8_t *d2, const uint8_t *good2, const int *node2, float nnratio, int check_ori, int *match12)
{
    int *m2 = (int *)malloc(sizeof(int) * (size_t)(n2 > 0 ? n2 : 1));
    const int n = search_by_bow_core(n1, k1, d1, good1, node1, n2, k2, d2, good2, node2, nnratio, check_ori, 1, m2);
    for (int i = 0; i < n1; i++) match12[i] = -1;
    for (int j = 0; j < n2; j++) if (m2[j] >= 0) match12[m2[j]] = j;
    free(m2);
    return n;
}

/* The search of ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th), ORBmatcher.cc:829-979: for every candidate map point the keyframe
 * keypoint it would be fused with (best_idx, -1 = none within TH_LOW) and its descriptor distance.  m_skip[i] = !pMP || isBad() || IsInKeyFrame(pKF).  The map mutations
 * that follow (Replace / AddObservation / AddMapPoint, :950-969) stay with the caller; nFused = number of best_idx >= 0. */
int orc_fuse_search(int Nk, const orc_keypoint *keys, const uint8_t *desc, const float *uright, const float *Tcw,
                    int Nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
                    float fx, float fy, float cx, float cy, float bf, float minX, float maxX, float minY, float maxY,
                    const float *scale_factors, const float *inv_level_sigma2, int nlevels, float log_scale_factor, float th, int *best_idx, int *best_dist)
{
    grid_t *g = (grid_t *)malloc(sizeof(grid_t));
    grid_build(g, Nk, keys, minX, maxX, minY, maxY);
    int *cand = (int *)malloc(sizeof(int) * (Nk > 0 ? Nk : 1));
    float Rcw[3][3], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r][c] = Tcw[4 * r + c]; tcw[r] = Tcw[4 * r + 3]; }
    for (int r = 0; r < 3; r++) Ow[r] = -(Rcw[0][r] * tcw[0] + Rcw[1][r] * tcw[1] + Rcw[2][r] * tcw[2]);       /* GetCameraCenter: -Rcw^T tcw (cv::Mat product) */
    int nFused = 0;
    for (int i = 0; i < Nm; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        if (m_skip[i]) continue;
        const float *P = m_xw + 3 * i;
        const float pcx = gemm3(Rcw[0], P, 1.0, 1.0, tcw[0]), pcy = gemm3(Rcw[1], P, 1.0, 1.0, tcw[1]), pcz = gemm3(Rcw[2], P, 1.0, 1.0, tcw[2]);
        if (pcz < 0.0f) continue;
        const float invz = 1 / pcz, x = pcx * invz, y = pcy * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= minX && u < maxX && v >= minY && v < maxY)) continue;                      /* KeyFrame::IsInImage */
        const float ur = u - bf * invz;
        const float maxDistance = 1.2f * m_max_dist[i], minDistance = 0.8f * m_min_dist[i];
        const float po0 = P[0] - Ow[0], po1 = P[1] - Ow[1], po2 = P[2] - Ow[2];
        const float dist3D = (float)sqrt((double)po0 * po0 + (double)po1 * po1 + (double)po2 * po2);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const double dot = (double)po0 * m_normal[3 * i] + (double)po1 * m_normal[3 * i + 1] + (double)po2 * m_normal[3 * i + 2];
        if (dot < 0.5 * dist3D) continue;
        int lvl = (int)ceilf(logf(m_max_dist[i] / dist3D) / log_scale_factor);
        if (lvl < 0) lvl = 0; else if (lvl >= nlevels) lvl = nlevels - 1;
        const float radius = th * scale_factors[lvl];
        const int nc = features_in_area(g, keys, u, v, radius, -1, -1, cand);
        int bestDist = 256, bestIdx = -1;
        for (int q = 0; q < nc; q++) {
            const int idx = cand[q]; const orc_keypoint *kp = &keys[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            if (uright[idx] >= 0) {
                const float ex = u - kp->x, ey = v - kp->y, er = ur - uright[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp->x, ey = v - kp->y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = orc_descriptor_distance(m_desc + 32 * (size_t)i, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; best_dist[i] = bestDist; nFused++; }
    }
    grid_free(g); free(g); free(cand);
    return nFused;
}

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist)
 * (ORBmatcher.cc:1474-1601; caller Tracking::Relocalization, Tracking.cc:1571,1584).  kf_ok[i] = vpMPs[i] && !isBad() && !sAlreadyFound.count();
 * c_has_mp[k] = CurrentFrame.mvpMapPoints[k] != NULL on entry.  cur_match[k] (out) = index i of the keyframe map point the frame's keypoint k receives, -1 = untouched. */
int orc_search_by_projection_kf(
    int Nc, const orc_keypoint *ckeys, const uint8_t *cdesc, const uint8_t *c_has_mp, const float *cTcw,
    int Nk, const orc_keypoint *kf_keys, const uint8_t *kf_ok, const float *m_xw, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc,
    float fx, float fy, float cx, float cy, float minX, float maxX, float minY, float maxY,
    const float *scale_factors, int nlevels, float log_scale_factor, float th, int orb_dist, int check_ori, int *cur_match)
{
    int nmatches = 0;
    grid_t *g = (grid_t *)malloc(sizeof(grid_t));
    grid_build(g, Nc, ckeys, minX, maxX, minY, maxY);
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH], hc[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { hist[i] = NULL; hn[i] = 0; hc[i] = 0; }
    const float factor = HISTO_LENGTH / 360.0f;
    float Rcw[3][3], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r][c] = cTcw[4 * r + c]; tcw[r] = cTcw[4 * r + 3]; }
    for (int i = 0; i < 3; i++) {                       /* Ow = -Rcw.t()*tcw : gemm with a transpose flag -> generic path, double accumulation, alpha = -1 */
        double s = 0; for (int k = 0; k < 3; k++) s += (double)Rcw[k][i] * (double)tcw[k];
        Ow[i] = (float)(s * -1.0);
    }
    for (int k = 0; k < Nc; k++) cur_match[k] = -1;
    uint8_t *taken = (uint8_t *)malloc(Nc > 0 ? Nc : 1);
    for (int k = 0; k < Nc; k++) taken[k] = c_has_mp[k] ? 1 : 0;
    int *vind = (int *)malloc(sizeof(int) * (Nc > 0 ? Nc : 1));
    for (int i = 0; i < Nk; i++) {
        if (!kf_ok[i]) continue;
        const float *xw = m_xw + 3 * i;
        const float xc = gemm3(Rcw[0], xw, 1.0, 1.0, tcw[0]), yc = gemm3(Rcw[1], xw, 1.0, 1.0, tcw[1]), zc = gemm3(Rcw[2], xw, 1.0, 1.0, tcw[2]);
        const float invzc = (float)(1.0 / zc);
        const float u = fx * xc * invzc + cx, v = fy * yc * invzc + cy;
        if (u < minX || u > maxX) continue;
        if (v < minY || v > maxY) continue;
        const float po0 = xw[0] - Ow[0], po1 = xw[1] - Ow[1], po2 = xw[2] - Ow[2];
        const float dist3D = (float)sqrt((double)po0 * po0 + (double)po1 * po1 + (double)po2 * po2);      /* cv::norm: double accumulation */
        const float maxDistance = 1.2f * m_max_dist[i], minDistance = 0.8f * m_min_dist[i];             /* Get{Max,Min}DistanceInvariance */
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        int lvl = (int)ceilf(logf(m_max_dist[i] / dist3D) / log_scale_factor);                             /* MapPoint::PredictScale(dist, &CurrentFrame) */
        if (lvl < 0) lvl = 0; else if (lvl >= nlevels) lvl = nlevels - 1;
        const float radius = th * scale_factors[lvl];
        const int nv = features_in_area(g, ckeys, u, v, radius, lvl - 1, lvl + 1, vind);
        if (nv == 0) continue;
        const uint8_t *dMP = m_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (int q = 0; q < nv; q++) {
            const int i2 = vind[q];
            if (taken[i2]) continue;                                                                       /* CurrentFrame.mvpMapPoints[i2] != NULL */
            const int dist = orc_descriptor_distance(dMP, cdesc + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {
            cur_match[bestIdx2] = i; taken[bestIdx2] = 1;
            nmatches++;
            if (check_ori) {
                float rot = kf_keys[i].angle - ckeys[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (hn[bin] == hc[bin]) { hc[bin] = hc[bin] ? 2 * hc[bin] : 64; hist[bin] = (int *)realloc(hist[bin], sizeof(int) * hc[bin]); }
                hist[bin][hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != i1 && i != i2 && i != i3)
                for (int j = 0; j < hn[i]; j++) { cur_match[hist[i][j]] = -1; nmatches--; }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    free(vind); free(taken); grid_free(g); free(g);
    return nmatches;
}

/* ---- loop-closing matchers that project through a Sim3 (LoopClosing::ComputeSim3 / SearchAndFuse) ---------------------------------------------------------------- */

/* "Decompose Scw" (ORBmatcher.cc:301-306, :990-995): scw = sqrt(row0 . row0) (Mat::dot: double accumulation), Rcw = sRcw / scw and tcw = Scw.col(3) / scw
 * (Mat / double -> convertTo with the float of 1/scw), Ow = -Rcw.t() * tcw (gemm with a transpose flag: double accumulation, alpha = -1). */
static void decompose_scw(const float *Scw, float Rcw[3][3], float tcw[3], float Ow[3])
{
    double s = 0; for (int c = 0; c < 3; c++) s += (double)Scw[c] * (double)Scw[c];
    const float scw = (float)sqrt(s);
    const float inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r][c] = Scw[4 * r + c] * inv; tcw[r] = Scw[4 * r + 3] * inv; }
    for (int i = 0; i < 3; i++) {
        double a = 0; for (int k = 0; k < 3; k++) a += (double)Rcw[k][i] * (double)tcw[k];
        Ow[i] = (float)(a * -1.0);
    }
}

/* The search of int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint) (ORBmatcher.cc:981-1101;
 * caller LoopClosing::SearchAndFuse, LoopClosing.cc:599).  m_skip[i] = isBad() || pKF->GetMapPoints().count(pMP).  best_idx[i] = keyframe keypoint the point lands on
 * (-1: none within TH_LOW); the caller then reads pKF->GetMapPoint(best_idx[i]) to fill vpReplacePoint or to add the observation (:1084-1097). */
int orc_fuse_search_sim3(int Nk, const orc_keypoint *keys, const uint8_t *desc, const float *Scw,
                         int Nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
                         float fx, float fy, float cx, float cy, float minX, float maxX, float minY, float maxY,
                         const float *scale_factors, int nlevels, float log_scale_factor, float th, int *best_idx, int *best_dist)
{
    grid_t *g = (grid_t *)malloc(sizeof(grid_t));
    grid_build(g, Nk, keys, minX, maxX, minY, maxY);
    int *cand = (int *)malloc(sizeof(int) * (Nk > 0 ? Nk : 1));
    float Rcw[3][3], tcw[3], Ow[3];
    decompose_scw(Scw, Rcw, tcw, Ow);
    int nFused = 0;
    for (int i = 0; i < Nm; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        if (m_skip[i]) continue;
        const float *P = m_xw + 3 * i;
        const float pcx = gemm3(Rcw[0], P, 1.0, 1.0, tcw[0]), pcy = gemm3(Rcw[1], P, 1.0, 1.0, tcw[1]), pcz = gemm3(Rcw[2], P, 1.0, 1.0, tcw[2]);
        if (pcz < 0.0f) continue;
        const float invz = (float)(1.0 / pcz), x = pcx * invz, y = pcy * invz;                 /* :1022 `1.0/` : double division */
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= minX && u < maxX && v >= minY && v < maxY)) continue;
        const float maxDistance = 1.2f * m_max_dist[i], minDistance = 0.8f * m_min_dist[i];
        const float po0 = P[0] - Ow[0], po1 = P[1] - Ow[1], po2 = P[2] - Ow[2];
        const float dist3D = (float)sqrt((double)po0 * po0 + (double)po1 * po1 + (double)po2 * po2);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const double dot = (double)po0 * m_normal[3 * i] + (double)po1 * m_normal[3 * i + 1] + (double)po2 * m_normal[3 * i + 2];
        if (dot < 0.5 * dist3D) continue;
        int lvl = (int)ceilf(logf(m_max_dist[i] / dist3D) / log_scale_factor);
        if (lvl < 0) lvl = 0; else if (lvl >= nlevels) lvl = nlevels - 1;
        const float radius = th * scale_factors[lvl];
        const int nc = features_in_area(g, keys, u, v, radius, -1, -1, cand);
        int bestDist = INT_MAX, bestIdx = -1;
        for (int q = 0; q < nc; q++) {
            const int idx = cand[q];
            const int kpLevel = keys[idx].octave;
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            const int dist = orc_descriptor_distance(m_desc + 32 * (size_t)i, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; best_dist[i] = bestDist; nFused++; }
    }
    grid_free(g); free(g); free(cand);
    return nFused;
}

/* int ORBmatcher::SearchByProjection(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th) (ORBmatcher.cc:292-407;
 * caller LoopClosing::ComputeSim3, LoopClosing.cc:375).  matched_in[k] = vpMatched[k] != NULL on entry; m_skip[i] = isBad() || the point is already in vpMatched;
 * matched_out[k] = index i of the candidate that this call put into vpMatched[k], -1 = unchanged. */
int orc_search_by_projection_sim3(int Nk, const orc_keypoint *keys, const uint8_t *desc, const uint8_t *matched_in, const float *Scw,
                                  int Nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
                                  float fx, float fy, float cx, float cy, float minX, float maxX, float minY, float maxY,
                                  const float *scale_factors, int nlevels, float log_scale_factor, int th, int *matched_out)
{
    grid_t *g = (grid_t *)malloc(sizeof(grid_t));
    grid_build(g, Nk, keys, minX, maxX, minY, maxY);
    int *cand = (int *)malloc(sizeof(int) * (Nk > 0 ? Nk : 1));
    uint8_t *taken = (uint8_t *)malloc((size_t)(Nk > 0 ? Nk : 1));
    for (int k = 0; k < Nk; k++) { taken[k] = matched_in[k] != 0; matched_out[k] = -1; }
    float Rcw[3][3], tcw[3], Ow[3];
    decompose_scw(Scw, Rcw, tcw, Ow);
    int nmatches = 0;
    for (int i = 0; i < Nm; i++) {
        if (m_skip[i]) continue;
        const float *P = m_xw + 3 * i;
        const float pcx = gemm3(Rcw[0], P, 1.0, 1.0, tcw[0]), pcy = gemm3(Rcw[1], P, 1.0, 1.0, tcw[1]), pcz = gemm3(Rcw[2], P, 1.0, 1.0, tcw[2]);
        if (pcz < 0.0f) continue;
        const float invz = 1 / pcz, x = pcx * invz, y = pcy * invz;                            /* :333 `1/` : float division */
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= minX && u < maxX && v >= minY && v < maxY)) continue;
        const float maxDistance = 1.2f * m_max_dist[i], minDistance = 0.8f * m_min_dist[i];
        const float po0 = P[0] - Ow[0], po1 = P[1] - Ow[1], po2 = P[2] - Ow[2];
        const float dist = (float)sqrt((double)po0 * po0 + (double)po1 * po1 + (double)po2 * po2);
        if (dist < minDistance || dist > maxDistance) continue;
        const double dot = (double)po0 * m_normal[3 * i] + (double)po1 * m_normal[3 * i + 1] + (double)po2 * m_normal[3 * i + 2];
        if (dot < 0.5 * dist) continue;
        int lvl = (int)ceilf(logf(m_max_dist[i] / dist) / log_scale_factor);
        if (lvl < 0) lvl = 0; else if (lvl >= nlevels) lvl = nlevels - 1;
        const float radius = th * scale_factors[lvl];
        const int nc = features_in_area(g, keys, u, v, radius, -1, -1, cand);
        int bestDist = 256, bestIdx = -1;
        for (int q = 0; q < nc; q++) {
            const int idx = cand[q];
            if (taken[idx]) continue;                                                          /* vpMatched[idx] */
            const int kpLevel = keys[idx].octave;
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            const int d = orc_descriptor_distance(m_desc + 32 * (size_t)i, desc + 32 * (size_t)idx);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { taken[bestIdx] = 1; matched_out[bestIdx] = i; nmatches++; }
    }
    grid_free(g); free(g); free(cand); free(taken);
    return nmatches;
}

/* one direction of SearchBySim3 (:1150-1224 / :1227-1301): map points of the source keyframe (indexed by its keypoints) through Rsw, tsw and then sR, t into the
 * target keyframe; match[i] = best target keypoint within TH_HIGH, -1 otherwise */
static void sim3_direction(int Ns, const uint8_t *src_ok, const uint8_t *src_already, const float *xw, const float *mind, const float *maxd, const uint8_t *mdesc,
                           const float Rsw[3][3], const float *tsw, const float sR[3][3], const float *t,
                           const grid_t *g, const orc_keypoint *tkeys, const uint8_t *tdesc, int *cand,
                           float fx, float fy, float cx, float cy, float minX, float maxX, float minY, float maxY,
                           const float *scale_factors, int nlevels, float log_scale_factor, float th, int *match)
{
    for (int i = 0; i < Ns; i++) {
        match[i] = -1;
        if (!src_ok[i] || src_already[i]) continue;
        const float *P = xw + 3 * i;
        const float c1[3] = { gemm3(Rsw[0], P, 1.0, 1.0, tsw[0]), gemm3(Rsw[1], P, 1.0, 1.0, tsw[1]), gemm3(Rsw[2], P, 1.0, 1.0, tsw[2]) };
        const float c2[3] = { gemm3(sR[0], c1, 1.0, 1.0, t[0]), gemm3(sR[1], c1, 1.0, 1.0, t[1]), gemm3(sR[2], c1, 1.0, 1.0, t[2]) };
        if (c2[2] < 0.0) continue;
        const float invz = (float)(1.0 / c2[2]), x = c2[0] * invz, y = c2[1] * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!(u >= minX && u < maxX && v >= minY && v < maxY)) continue;
        const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i];
        const float dist3D = (float)sqrt((double)c2[0] * c2[0] + (double)c2[1] * c2[1] + (double)c2[2] * c2[2]);       /* cv::norm(p3Dc2) */
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        int lvl = (int)ceilf(logf(maxd[i] / dist3D) / log_scale_factor);
        if (lvl < 0) lvl = 0; else if (lvl >= nlevels) lvl = nlevels - 1;
        const float radius = th * scale_factors[lvl];
        const int nc = features_in_area(g, tkeys, u, v, radius, -1, -1, cand);
        int bestDist = INT_MAX, bestIdx = -1;
        for (int q = 0; q < nc; q++) {
            const int idx = cand[q];
            if (tkeys[idx].octave < lvl - 1 || tkeys[idx].octave > lvl) continue;
            const int d = orc_descriptor_distance(mdesc + 32 * (size_t)i, tdesc + 32 * (size_t)idx);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_HIGH) match[i] = bestIdx;
    }
}

/* int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th)
 * (ORBmatcher.cc:1106-1330; caller LoopClosing::ComputeSim3, LoopClosing.cc:323).  Per keyframe, indexed by keypoint: ok = holds a map point that is not bad, xw / min_dist /
 * max_dist / mdesc = that map point's position, mfMinDistance, mfMaxDistance and descriptor.  match12 (in/out): -1 = vpMatches12[i1] is NULL, >= 0 = the keypoint of pKF2
 * its map point sits on (GetIndexInKeyFrame), -2 = non-NULL but not observed by pKF2.  New matches are written as keypoint indices of pKF2; returns nFound. */
int orc_search_by_sim3(int N1, const orc_keypoint *keys1, const uint8_t *desc1, const float *Tcw1, const uint8_t *ok1, const float *xw1, const float *min1, const float *max1, const uint8_t *mdesc1,
                       int N2, const orc_keypoint *keys2, const uint8_t *desc2, const float *Tcw2, const uint8_t *ok2, const float *xw2, const float *min2, const float *max2, const uint8_t *mdesc2,
                       float fx, float fy, float cx, float cy, float minX, float maxX, float minY, float maxY,
                       const float *scale_factors, int nlevels, float log_scale_factor, float s12, const float *R12, const float *t12, float th, int *match12)
{
    float R1w[3][3], t1w[3], R2w[3][3], t2w[3], sR12[3][3], sR21[3][3], t21[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { R1w[r][c] = Tcw1[4 * r + c]; R2w[r][c] = Tcw2[4 * r + c]; } t1w[r] = Tcw1[4 * r + 3]; t2w[r] = Tcw2[4 * r + 3]; }
    const float inv_s = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sR12[r][c] = R12[3 * r + c] * s12; sR21[r][c] = R12[3 * c + r] * inv_s; }     /* s12*R12, (1.0/s12)*R12.t() */
    for (int r = 0; r < 3; r++) t21[r] = gemm3(sR21[r], t12, -1.0, 0.0, 0.0f);                                                             /* -sR21*t12 */
    uint8_t *already1 = (uint8_t *)calloc((size_t)(N1 > 0 ? N1 : 1), 1), *already2 = (uint8_t *)calloc((size_t)(N2 > 0 ? N2 : 1), 1);
    for (int i = 0; i < N1; i++) if (match12[i] != -1) { already1[i] = 1; if (match12[i] >= 0 && match12[i] < N2) already2[match12[i]] = 1; }
    int *m1 = (int *)malloc(sizeof(int) * (size_t)(N1 > 0 ? N1 : 1)), *m2 = (int *)malloc(sizeof(int) * (size_t)(N2 > 0 ? N2 : 1));
    int *cand = (int *)malloc(sizeof(int) * (size_t)((N1 > N2 ? N1 : N2) + 1));
    grid_t *g = (grid_t *)malloc(sizeof(grid_t));
    grid_build(g, N2, keys2, minX, maxX, minY, maxY);
    sim3_direction(N1, ok1, already1, xw1, min1, max1, mdesc1, R1w, t1w, sR21, t21, g, keys2, desc2, cand, fx, fy, cx, cy, minX, maxX, minY, maxY, scale_factors, nlevels, log_scale_factor, th, m1);
    grid_free(g);
    grid_build(g, N1, keys1, minX, maxX, minY, maxY);
    sim3_direction(N2, ok2, already2, xw2, min2, max2, mdesc2, R2w, t2w, sR12, t12, g, keys1, desc1, cand, fx, fy, cx, cy, minX, maxX, minY, maxY, scale_factors, nlevels, log_scale_factor, th, m2);
    grid_free(g); free(g);
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {                                                   /* check agreement, :1305-1320 */
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nFound++; }
    }
    free(already1); free(already2); free(m1); free(m2); free(cand);
    return nFound;
}

/* int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)
 * (ORBmatcher.cc:407-522; caller Tracking::MonocularInitialization, Tracking.cc:639 — the monocular initialiser, never reached by the RGB-D system).
 * prev_matched (in/out, N1 x 2 floats) = vbPrevMatched; matches12 (out, N1) = vnMatches12.  A histogram entry stays in its bin when a later keypoint steals the match
 * (the three maxima count it; the removal loop skips it because vnMatches12 is already -1), as in the reference. */
int orc_search_for_initialization(int N1, const orc_keypoint *k1, const uint8_t *d1, int N2, const orc_keypoint *k2, const uint8_t *d2, float *prev_matched, int windowSize,
                                  float nnratio, int check_ori, float minX, float maxX, float minY, float maxY, int *matches12)
{
    int nmatches = 0;
    grid_t *g = (grid_t *)malloc(sizeof(grid_t));
    grid_build(g, N2, k2, minX, maxX, minY, maxY);
    int *vind = (int *)malloc(sizeof(int) * (size_t)(N2 > 0 ? N2 : 1)), *dist2 = (int *)malloc(sizeof(int) * (size_t)(N2 > 0 ? N2 : 1)), *m21 = (int *)malloc(sizeof(int) * (size_t)(N2 > 0 ? N2 : 1));
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH], hc[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { hist[i] = NULL; hn[i] = 0; hc[i] = 0; }
    const float factor = HISTO_LENGTH / 360.0f;
    for (int i = 0; i < N1; i++) matches12[i] = -1;
    for (int i = 0; i < N2; i++) { dist2[i] = INT_MAX; m21[i] = -1; }
    for (int i1 = 0; i1 < N1; i1++) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        const int nv = features_in_area(g, k2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)windowSize, level1, level1, vind);
        if (nv == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int q = 0; q < nv; q++) {
            const int i2 = vind[q];
            const int dist = orc_descriptor_distance(d1 + 32 * (size_t)i1, d2 + 32 * (size_t)i2);
            if (dist2[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (m21[bestIdx2] >= 0) { matches12[m21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2; m21[bestIdx2] = i1; dist2[bestIdx2] = bestDist; nmatches++;
                if (check_ori) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    if (hn[bin] == hc[bin]) { hc[bin] = hc[bin] ? 2 * hc[bin] : 64; hist[bin] = (int *)realloc(hist[bin], sizeof(int) * hc[bin]); }
                    hist[bin][hn[bin]++] = i1;
                }
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; j++) { const int idx1 = hist[i][j]; if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; } }
        }
    }
    for (int i = 0; i < N1; i++) if (matches12[i] >= 0) { prev_matched[2 * i] = k2[matches12[i]].x; prev_matched[2 * i + 1] = k2[matches12[i]].y; }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    grid_free(g); free(g); free(vind); free(dist2); free(m21);
    return nmatches;
}

/* ---- MapPoint post-steps of the optimisers and of map-point creation ---------------------------------------------------------------------------------------------
 * void MapPoint::UpdateNormalAndDepth() (src/sg-slam/src/MapPoint.cc:330-371; callers Optimizer.cc:227,776,1042, LocalMapping.cc:152,444,527, Tracking.cc:572,708,1234) for n
 * points: obs_start = CSR over the observations of every point IN THE ORDER THE REFERENCE WALKS mObservations (a std::map keyed by KeyFrame*: pointer order — the caller
 * supplies it), obs_center = GetCameraCenter() of each observing keyframe, ref_center / ref_level = the reference keyframe's camera centre and the octave of the point's
 * keypoint in it.  A point without observations keeps its values (the function returns early, :345-346). */
void orc_update_normal_and_depth(int n, const float *xw, const int *obs_start, const float *obs_center, const float *ref_center, const int *ref_level,
                                 const float *scale_factors, int nlevels, float *normal, float *min_dist, float *max_dist)
{
    for (int p = 0; p < n; p++) {
        const int s = obs_start[p], e = obs_start[p + 1];
        if (e <= s) continue;
        const float *P = xw + 3 * p;
        float nrm[3] = { 0.f, 0.f, 0.f };
        for (int q = s; q < e; q++) {
            const float d[3] = { P[0] - obs_center[3 * q], P[1] - obs_center[3 * q + 1], P[2] - obs_center[3 * q + 2] };          /* normali = mWorldPos - Owi */
            const double len = sqrt((double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2]);                                /* cv::norm */
            const float inv = (float)(1.0 / len);                                                                                   /* normali / norm: convertTo by the float of 1/norm */
            for (int k = 0; k < 3; k++) nrm[k] = nrm[k] + d[k] * inv;
        }
        const float pc[3] = { P[0] - ref_center[3 * p], P[1] - ref_center[3 * p + 1], P[2] - ref_center[3 * p + 2] };
        const float dist = (float)sqrt((double)pc[0] * pc[0] + (double)pc[1] * pc[1] + (double)pc[2] * pc[2]);
        max_dist[p] = dist * scale_factors[ref_level[p]];
        min_dist[p] = max_dist[p] / scale_factors[nlevels - 1];
        const float invn = (float)(1.0 / (double)(e - s));                                                                          /* normal / n */
        for (int k = 0; k < 3; k++) normal[3 * p + k] = nrm[k] * invn;
    }
}

/* void MapPoint::ComputeDistinctiveDescriptors() (MapPoint.cc:242-307) for n points: obs_desc = the observed descriptor rows (keyframes that are not bad), in mObservations
 * order.  best[p] = index (within the point's list) of the descriptor with the least median distance to the others, -1 for an empty list (early return, :269-270). */
void orc_distinctive_descriptors(int n, const int *obs_start, const uint8_t *obs_desc, int *best)
{
    for (int p = 0; p < n; p++) {
        const int s = obs_start[p], N = obs_start[p + 1] - s;
        best[p] = -1;
        if (N <= 0) continue;
        int *row = (int *)malloc(sizeof(int) * (size_t)N);
        int BestMedian = INT_MAX, BestIdx = 0;
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < N; j++) row[j] = i == j ? 0 : orc_descriptor_distance(obs_desc + 32 * (size_t)(s + i), obs_desc + 32 * (size_t)(s + j));
            for (int a = 1; a < N; a++) { const int v = row[a]; int b = a - 1; while (b >= 0 && row[b] > v) { row[b + 1] = row[b]; b--; } row[b + 1] = v; }        /* sort */
            const int median = row[(int)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best[p] = BestIdx;
        free(row);
    }
}

/* ---- the triangulation step of LocalMapping::CreateNewMapPoints (src/sg-slam/src/LocalMapping.cc:283-421) for the pairs SearchForTriangulation returned -----------
 * cv::SVD::compute(A, w, u, vt, MODIFY_A | FULL_UV) on the 4 x 4 float matrix = OpenCV's one-sided Jacobi (JacobiSVDImpl_<float> on A^T, lapack.cpp) when OpenCV is built
 * without LAPACK: rows of A^T are rotated pairwise until orthogonal, singular values sorted descending, vt.row(3) = the direction of the smallest one. */
static void jacobi_svd4_vt(const float *A /* 4x4 row-major */, float *vt_row3)
{
    const int n = 4, m = 4; const float eps = FLT_EPSILON * 2;       /* JacobiSVD(float*): JacobiSVDImpl_(..., FLT_MIN, FLT_EPSILON * 2) */
    float At[16], Vt[16]; double W[4];
    for (int i = 0; i < 4; i++) for (int k = 0; k < 4; k++) At[4 * i + k] = A[4 * k + i];
    for (int i = 0; i < n; i++) { double sd = 0; for (int k = 0; k < m; k++) { const float t = At[4 * i + k]; sd += (double)t * t; } W[i] = sd; for (int k = 0; k < n; k++) Vt[4 * i + k] = i == k ? 1.f : 0.f; }
    const int max_iter = 30;                                         /* max(m, 30) */
    for (int iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                float *Ai = At + 4 * i, *Aj = At + 4 * j;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot(p, beta);
                float c, s;
                if (beta < 0) { const double delta = (gamma - beta) * 0.5; s = (float)sqrt(delta / gamma); c = (float)(p / (gamma * s * 2)); }
                else { c = (float)sqrt((gamma + beta) / (gamma * 2)); s = (float)(p / (gamma * c * 2)); }
                a = b = 0;
                for (int k = 0; k < m; k++) { const float t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k]; Ai[k] = t0; Aj[k] = t1; a += (double)t0 * t0; b += (double)t1 * t1; }
                W[i] = a; W[j] = b; changed = 1;
                float *Vi = Vt + 4 * i, *Vj = Vt + 4 * j;
                for (int k = 0; k < n; k++) { const float t0 = c * Vi[k] + s * Vj[k], t1 = -s * Vi[k] + c * Vj[k]; Vi[k] = t0; Vj[k] = t1; }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) { double sd = 0; for (int k = 0; k < m; k++) { const float t = At[4 * i + k]; sd += (double)t * t; } W[i] = sqrt(sd); }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) { const double tw = W[i]; W[i] = W[j]; W[j] = tw; for (int k = 0; k < 4; k++) { float t = At[4 * i + k]; At[4 * i + k] = At[4 * j + k]; At[4 * j + k] = t; t = Vt[4 * i + k]; Vt[4 * i + k] = Vt[4 * j + k]; Vt[4 * j + k] = t; } }
    }
    for (int k = 0; k < 4; k++) vt_row3[k] = Vt[12 + k];
}

static double dot3d(const float *a, const float *b) { double r = 0; for (int i = 0; i < 3; i++) r += (double)a[i] * b[i]; return r; }      /* Mat::dot, 3 elements */

/* keys*_un = mvKeysUn, keys* = mvKeys (KeyFrame::UnprojectStereo reads the distorted ones, KeyFrame.cc:621-622), uright = mvuRight, depth = mvDepth; cam = fx, fy, cx, cy, mbf
 * (invfx = 1.0f / fx, mb = mbf / fx as Frame.cc:117-123 sets them).  ok[i] = 1 and x3d[i] = the new map point's position when pair i passes every gate. */
int orc_triangulate_pairs(int npairs, const int *pairs, const orc_keypoint *k1un, const orc_keypoint *k1, const float *ur1, const float *dp1, const float *Tcw1,
                          const orc_keypoint *k2un, const orc_keypoint *k2, const float *ur2, const float *dp2, const float *Tcw2,
                          float fx, float fy, float cx, float cy, float mbf, const float *scale_factors, const float *level_sigma2, float scale_factor, uint8_t *ok, float *x3d)
{
    const float invfx = 1.0f / fx, invfy = 1.0f / fy, mb = mbf / fx;
    float Rcw1[3][3], tcw1[3], Rwc1[3][3], Ow1[3], Rcw2[3][3], tcw2[3], Rwc2[3][3], Ow2[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { Rcw1[r][c] = Tcw1[4 * r + c]; Rcw2[r][c] = Tcw2[4 * r + c]; Rwc1[c][r] = Tcw1[4 * r + c]; Rwc2[c][r] = Tcw2[4 * r + c]; } tcw1[r] = Tcw1[4 * r + 3]; tcw2[r] = Tcw2[4 * r + 3]; }
    for (int i = 0; i < 3; i++) {                                    /* Ow = -Rcw.t() * tcw (KeyFrame::SetPose): transpose flag -> double accumulation, alpha = -1 */
        double a = 0, b = 0; for (int k = 0; k < 3; k++) { a += (double)Rcw1[k][i] * tcw1[k]; b += (double)Rcw2[k][i] * tcw2[k]; }
        Ow1[i] = (float)(a * -1.0); Ow2[i] = (float)(b * -1.0);
    }
    const float ratioFactor = 1.5f * scale_factor;
    int nnew = 0;
    for (int q = 0; q < npairs; q++) {
        ok[q] = 0; x3d[3 * q] = x3d[3 * q + 1] = x3d[3 * q + 2] = 0;
        const int idx1 = pairs[2 * q], idx2 = pairs[2 * q + 1];
        const orc_keypoint *kp1 = &k1un[idx1], *kp2 = &k2un[idx2];
        const float kp1_ur = ur1[idx1], kp2_ur = ur2[idx2];
        const int bStereo1 = kp1_ur >= 0, bStereo2 = kp2_ur >= 0;
        const float xn1[3] = { (kp1->x - cx) * invfx, (kp1->y - cy) * invfy, 1.0f }, xn2[3] = { (kp2->x - cx) * invfx, (kp2->y - cy) * invfy, 1.0f };
        const float ray1[3] = { gemm3(Rwc1[0], xn1, 1.0, 0.0, 0.f), gemm3(Rwc1[1], xn1, 1.0, 0.0, 0.f), gemm3(Rwc1[2], xn1, 1.0, 0.0, 0.f) };
        const float ray2[3] = { gemm3(Rwc2[0], xn2, 1.0, 0.0, 0.f), gemm3(Rwc2[1], xn2, 1.0, 0.0, 0.f), gemm3(Rwc2[2], xn2, 1.0, 0.0, 0.f) };
        const float cosParallaxRays = (float)(dot3d(ray1, ray2) / (sqrt(dot3d(ray1, ray1)) * sqrt(dot3d(ray2, ray2))));
        float cosParallaxStereo = cosParallaxRays + 1, cosParallaxStereo1 = cosParallaxStereo, cosParallaxStereo2 = cosParallaxStereo;
        if (bStereo1) cosParallaxStereo1 = cosf(2 * atan2f(mb / 2, dp1[idx1]));
        else if (bStereo2) cosParallaxStereo2 = cosf(2 * atan2f(mb / 2, dp2[idx2]));
        cosParallaxStereo = cosParallaxStereo1 < cosParallaxStereo2 ? cosParallaxStereo1 : cosParallaxStereo2;
        float X[3];
        if (cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && (bStereo1 || bStereo2 || cosParallaxRays < 0.9998)) {
            float A[16], v[4];
            for (int k = 0; k < 4; k++) {
                A[k] = xn1[0] * Tcw1[8 + k] - Tcw1[k]; A[4 + k] = xn1[1] * Tcw1[8 + k] - Tcw1[4 + k];
                A[8 + k] = xn2[0] * Tcw2[8 + k] - Tcw2[k]; A[12 + k] = xn2[1] * Tcw2[8 + k] - Tcw2[4 + k];
            }
            jacobi_svd4_vt(A, v);
            if (v[3] == 0) continue;
            const float inv = (float)(1.0 / (double)v[3]);
            X[0] = v[0] * inv; X[1] = v[1] * inv; X[2] = v[2] * inv;
        } else if (bStereo1 && cosParallaxStereo1 < cosParallaxStereo2) {
            const float z = dp1[idx1];
            if (!(z > 0)) continue;                                  /* UnprojectStereo returns an empty Mat: the reference would fault on x3D.t(); treated as rejected */
            const float xc[3] = { (k1[idx1].x - cx) * z * invfx, (k1[idx1].y - cy) * z * invfy, z };
            for (int i = 0; i < 3; i++) X[i] = gemm3(Rwc1[i], xc, 1.0, 1.0, Ow1[i]);
        } else if (bStereo2 && cosParallaxStereo2 < cosParallaxStereo1) {
            const float z = dp2[idx2];
            if (!(z > 0)) continue;
            const float xc[3] = { (k2[idx2].x - cx) * z * invfx, (k2[idx2].y - cy) * z * invfy, z };
            for (int i = 0; i < 3; i++) X[i] = gemm3(Rwc2[i], xc, 1.0, 1.0, Ow2[i]);
        } else continue;
        const float z1 = (float)(dot3d(Rcw1[2], X) + tcw1[2]);
        if (z1 <= 0) continue;
        const float z2 = (float)(dot3d(Rcw2[2], X) + tcw2[2]);
        if (z2 <= 0) continue;
        const float s1 = level_sigma2[kp1->octave];
        const float x1 = (float)(dot3d(Rcw1[0], X) + tcw1[0]), y1 = (float)(dot3d(Rcw1[1], X) + tcw1[1]);
        const float invz1 = (float)(1.0 / z1);
        {
            const float u1 = fx * x1 * invz1 + cx, v1 = fy * y1 * invz1 + cy, eX = u1 - kp1->x, eY = v1 - kp1->y;
            if (!bStereo1) { if ((eX * eX + eY * eY) > 5.991 * s1) continue; }
            else { const float u1r = u1 - mbf * invz1, eR = u1r - kp1_ur; if ((eX * eX + eY * eY + eR * eR) > 7.8 * s1) continue; }
        }
        const float s2 = level_sigma2[kp2->octave];
        const float x2 = (float)(dot3d(Rcw2[0], X) + tcw2[0]), y2 = (float)(dot3d(Rcw2[1], X) + tcw2[1]);
        const float invz2 = (float)(1.0 / z2);
        {
            const float u2 = fx * x2 * invz2 + cx, v2 = fy * y2 * invz2 + cy, eX = u2 - kp2->x, eY = v2 - kp2->y;
            if (!bStereo2) { if ((eX * eX + eY * eY) > 5.991 * s2) continue; }
            else { const float u2r = u2 - mbf * invz2, eR = u2r - kp2_ur; if ((eX * eX + eY * eY + eR * eR) > 7.8 * s2) continue; }
        }
        const float n1[3] = { X[0] - Ow1[0], X[1] - Ow1[1], X[2] - Ow1[2] }, n2[3] = { X[0] - Ow2[0], X[1] - Ow2[1], X[2] - Ow2[2] };
        const float dist1 = (float)sqrt(dot3d(n1, n1)), dist2 = (float)sqrt(dot3d(n2, n2));
        if (dist1 == 0 || dist2 == 0) continue;
        const float ratioDist = dist2 / dist1, ratioOctave = scale_factors[kp1->octave] / scale_factors[kp2->octave];
        if (ratioDist * ratioFactor < ratioOctave || ratioDist > ratioOctave * ratioFactor) continue;
        ok[q] = 1; x3d[3 * q] = X[0]; x3d[3 * q + 1] = X[1]; x3d[3 * q + 2] = X[2]; nnew++;
    }
    return nnew;
}
