// sgx_match2_kernels.h — ORBmatcher gates used by LocalMapping (tier N2 of SURVEY.md §8):
//   k_hamming_matrix            ORBmatcher::DescriptorDistance for every pair of two descriptor sets        src/sg-slam/src/ORBmatcher.cc:1649-1665
//   k_search_triangulation      ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)   src/sg-slam/src/ORBmatcher.cc:659-827
// SearchForTriangulation walks the two DBoW2 FeatureVectors (std::map<NodeId, vector<feature index>>) in lock step and, inside every COMMON vocabulary node,
// gives each unmatched keypoint of KF1 (index order) its best still-unmatched keypoint of KF2 of that node (Hamming <= TH_LOW, epipolar gate).  The greedy
// "vbMatched2" lock only couples keypoints of the SAME node, so nodes are independent: one thread runs one node's loops exactly as written, all nodes in
// parallel; the rotation histogram (ComputeThreeMaxima) is resolved by the workgroup afterwards.
#pragma once
#include "sgx_match_common.h"
#include <float.h>

#define SGX_TH_LOW 50             /* ORBmatcher::TH_LOW, ORBmatcher.cc:38 */

// out[i * nb + j] = Hamming distance of row i of A and row j of B (32-byte rows); 16 x 16 pairs per workgroup from LDS-staged rows
SGX_KERNEL(256) k_hamming_matrix(const uint32_t *A, int na, const uint32_t *B, int nb, uint16_t *out)
{
    SGX_LDS uint32_t sa[16][9], sb[16][9];
    const int i0 = (int)blockIdx.y * 16, j0 = (int)blockIdx.x * 16;
    SGX_THREADS_BEGIN(tid)
    const int r = tid >> 4, w = tid & 15;
    if (w < 8) sa[r][w] = i0 + r < na ? A[(size_t)(i0 + r) * 8 + w] : 0u;
    else sb[r][w - 8] = j0 + r < nb ? B[(size_t)(j0 + r) * 8 + (w - 8)] : 0u;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int i = tid >> 4, j = tid & 15;
    if (i0 + i < na && j0 + j < nb) out[(size_t)(i0 + i) * nb + j0 + j] = (uint16_t)sgx_hamming256(sa[i], sb[j]);
    SGX_THREADS_END
}

struct SgxTriArgs {
    int n1, n2, nnodes;                      /* keypoints of KF1 / KF2, common vocabulary nodes */
    const uint8_t *keys1, *keys2;            /* mvKeysUn (28-byte cv::KeyPoint) */
    const uint32_t *desc1, *desc2;
    const float *uright1, *uright2;
    const uint8_t *has_mp1, *has_mp2;        /* GetMapPoint(idx) != NULL */
    const int *items1, *items2;              /* feature indices grouped by node (FeatureVector order) */
    const int *job;                          /* per common node: start1, end1, start2, end2 into items1 / items2 */
    float F12[9];                            /* row-major */
    float ex, ey;                            /* epipole of KF1's centre in KF2 */
    SgxScales scale2, sigma2_2;              /* pKF2->mvScaleFactors, pKF2->mvLevelSigma2 */
    int only_stereo, check_ori;
    int *match12;                            /* out: n1 entries, index in KF2 or -1 */
    uint8_t *matched2;                       /* work: n2 flags */
    int *nmatches;
};

SGX_KERNEL(256) k_search_triangulation(SgxTriArgs A)
{
    SGX_LDS int hist[SGX_HISTO], bad[SGX_HISTO];
    SGX_LDS int s_total, s_rejected;
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < A.n1; i += 256) A.match12[i] = -1;
    for (int i = tid; i < A.n2; i += 256) A.matched2[i] = 0;
    for (int i = tid; i < SGX_HISTO; i += 256) { hist[i] = 0; bad[i] = 0; }
    if (tid == 0) { s_total = 0; s_rejected = 0; }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int nd = tid; nd < A.nnodes; nd += 256) {
        const int s1 = A.job[4 * nd], e1 = A.job[4 * nd + 1], s2 = A.job[4 * nd + 2], e2 = A.job[4 * nd + 3];
        for (int q1 = s1; q1 < e1; q1++) {
            const int idx1 = A.items1[q1];
            if (A.has_mp1[idx1]) continue;                                       // :704-705
            const bool stereo1 = A.uright1[idx1] >= 0;
            if (A.only_stereo && !stereo1) continue;
            const float *kp1 = (const float *)(A.keys1 + (size_t)idx1 * 28);
            const uint32_t *d1 = A.desc1 + (size_t)idx1 * 8;
            // CheckDistEpipolarLine: l = x1' F12 (ORBmatcher.cc:140-143), float arithmetic left to right
            const float a = kp1[0] * A.F12[0] + kp1[1] * A.F12[3] + A.F12[6];
            const float b = kp1[0] * A.F12[1] + kp1[1] * A.F12[4] + A.F12[7];
            const float c = kp1[0] * A.F12[2] + kp1[1] * A.F12[5] + A.F12[8];
            const float den = a * a + b * b;
            int bestDist = SGX_TH_LOW, bestIdx2 = -1;
            for (int q2 = s2; q2 < e2; q2++) {
                const int idx2 = A.items2[q2];
                if (A.matched2[idx2] || A.has_mp2[idx2]) continue;               // :728-729
                const bool stereo2 = A.uright2[idx2] >= 0;
                if (A.only_stereo && !stereo2) continue;
                const int dist = sgx_hamming256(d1, A.desc2 + (size_t)idx2 * 8);
                if (dist > SGX_TH_LOW || dist > bestDist) continue;
                const float *kp2 = (const float *)(A.keys2 + (size_t)idx2 * 28);
                const int oct2 = ((const int *)kp2)[5];
                if (!stereo1 && !stereo2) {                                      // too close to the epipole: :746-752
                    const float dx = A.ex - kp2[0], dy = A.ey - kp2[1];
                    if (dx * dx + dy * dy < 100 * A.scale2.s[oct2]) continue;
                }
                const float num = a * kp2[0] + b * kp2[1] + c;
                if (den == 0) continue;
                const float dsqr = num * num / den;
                if ((double)dsqr < 3.84 * (double)A.sigma2_2.s[oct2]) { bestIdx2 = idx2; bestDist = dist; }      // :157 compares in double (3.84 is a double literal)
            }
            if (bestIdx2 >= 0) {
                A.match12[idx1] = bestIdx2; A.matched2[bestIdx2] = 1;
                sgx_atomic_add(&s_total, 1);
                if (A.check_ori) {
                    const float *kp2 = (const float *)(A.keys2 + (size_t)bestIdx2 * 28);
                    float rot = kp1[3] - kp2[3];
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
                    if (bin == SGX_HISTO) bin = 0;
                    sgx_atomic_add(&hist[bin], 1);
                }
            }
        }
    }
    SGX_THREADS_END
    SGX_SYNC();
    if (A.check_ori) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {                         // ComputeThreeMaxima, ORBmatcher.cc:1603-1644
            int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < SGX_HISTO; i++) {
                const int s = hist[i];
                if (s > m1) { m3 = m2; m2 = m1; m1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > m2) { m3 = m2; m2 = s; i3 = i2; i2 = i; }
                else if (s > m3) { m3 = s; i3 = i; }
            }
            if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
            else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
            for (int i = 0; i < SGX_HISTO; i++) bad[i] = (i != i1 && i != i2 && i != i3);
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < A.n1; i += 256) {
            const int j = A.match12[i];
            if (j < 0) continue;
            const float *kp1 = (const float *)(A.keys1 + (size_t)i * 28), *kp2 = (const float *)(A.keys2 + (size_t)j * 28);
            float rot = kp1[3] - kp2[3];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
            if (bin == SGX_HISTO) bin = 0;
            if (bad[bin]) { A.match12[i] = -1; A.matched2[j] = 0; sgx_atomic_add(&s_rejected, 1); }
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) A.nmatches[0] = s_total - s_rejected;
    SGX_THREADS_END
}


// ---------------------------------------------------------------------------------------------
// k_search_bow: ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vpMapPointMatches)   src/sg-slam/src/ORBmatcher.cc:159-290.  Same node-by-node walk as
// SearchForTriangulation: inside a common vocabulary node every keyframe keypoint that holds a good map point (index order) takes the closest still-free frame
// keypoint of the node when it passes TH_LOW and the nearest-neighbour ratio; one thread per node, rotation histogram by the workgroup.
// ---------------------------------------------------------------------------------------------
struct SgxBowArgs {
    int nk, nf, nnodes;
    const uint8_t *keys_k, *keys_f; const uint32_t *desc_k, *desc_f; const uint8_t *good_k;
    const uint8_t *good_f;      // KeyFrame-KeyFrame overload: side 2 must hold a good map point too (ORBmatcher.cc:581-587); NULL = KeyFrame-Frame overload
    const int *items_k, *items_f, *job;
    float nnratio; int check_ori;
    int th_low;                 // accept bestDist1 <= th_low: TH_LOW for KeyFrame-Frame (:234), TH_LOW - 1 for KeyFrame-KeyFrame (`< TH_LOW`, :597)
    int *match_f; int *nmatches;
};
SGX_KERNEL(256) k_search_bow(SgxBowArgs A)
{
    SGX_LDS int hist[SGX_HISTO], bad[SGX_HISTO];
    SGX_LDS int s_total, s_rejected;
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < A.nf; i += 256) A.match_f[i] = -1;
    for (int i = tid; i < SGX_HISTO; i += 256) { hist[i] = 0; bad[i] = 0; }
    if (tid == 0) { s_total = 0; s_rejected = 0; }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int nd = tid; nd < A.nnodes; nd += 256) {
        const int s1 = A.job[4 * nd], e1 = A.job[4 * nd + 1], s2 = A.job[4 * nd + 2], e2 = A.job[4 * nd + 3];
        for (int q1 = s1; q1 < e1; q1++) {
            const int ik = A.items_k[q1];
            if (!A.good_k[ik]) continue;                                          // :195-199
            const uint32_t *d1 = A.desc_k + (size_t)ik * 8;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
            for (int q2 = s2; q2 < e2; q2++) {
                const int jf = A.items_f[q2];
                if (A.match_f[jf] >= 0) continue;                                  // :212 / vbMatched2 :578
                if (A.good_f && !A.good_f[jf]) continue;                           // :581-587
                const int dist = sgx_hamming256(d1, A.desc_f + (size_t)jf * 8);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = jf; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 <= A.th_low && (float)bestDist1 < A.nnratio * (float)bestDist2) {
                A.match_f[bestIdxF] = ik;
                sgx_atomic_add(&s_total, 1);
                if (A.check_ori) {
                    const float ka = ((const float *)(A.keys_k + (size_t)ik * 28))[3], fa = ((const float *)(A.keys_f + (size_t)bestIdxF * 28))[3];
                    float rot = ka - fa;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
                    if (bin == SGX_HISTO) bin = 0;
                    sgx_atomic_add(&hist[bin], 1);
                }
            }
        }
    }
    SGX_THREADS_END
    SGX_SYNC();
    if (A.check_ori) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {                         // ComputeThreeMaxima, ORBmatcher.cc:1603-1644
            int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < SGX_HISTO; i++) {
                const int s = hist[i];
                if (s > m1) { m3 = m2; m2 = m1; m1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > m2) { m3 = m2; m2 = s; i3 = i2; i2 = i; }
                else if (s > m3) { m3 = s; i3 = i; }
            }
            if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
            else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
            for (int i = 0; i < SGX_HISTO; i++) bad[i] = (i != i1 && i != i2 && i != i3);
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int j = tid; j < A.nf; j += 256) {
            const int ik = A.match_f[j];
            if (ik < 0) continue;
            const float ka = ((const float *)(A.keys_k + (size_t)ik * 28))[3], fa = ((const float *)(A.keys_f + (size_t)j * 28))[3];
            float rot = ka - fa;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
            if (bin == SGX_HISTO) bin = 0;
            if (bad[bin]) { A.match_f[j] = -1; sgx_atomic_add(&s_rejected, 1); }
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) A.nmatches[0] = s_total - s_rejected;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_fuse_search: the search of ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th)   src/sg-slam/src/ORBmatcher.cc:829-979
// One thread per candidate map point: projection + gates (:855-897), KeyFrame::GetFeaturesInArea over the keyframe's 64 x 48 grid in the reference's scan order
// (KeyFrame.cc:570-609), level / chi2 gates (:909-941), best Hamming distance with first-wins ties (:947-951).  No locks: candidates do not depend on earlier fusions.
// ---------------------------------------------------------------------------------------------
struct SgxFuseArgs {
    int nk, nm, nlevels;
    const uint8_t *keys; const uint32_t *desc; const float *uright;
    const int *cell_start, *cell_items;                 // CSR of the keyframe grid: cell (ix, iy) -> ix * 48 + iy
    const float *m_xw, *m_normal, *m_min_dist, *m_max_dist; const uint32_t *m_desc; const uint8_t *m_skip;
    float Rcw[3][3], tcw[3], Ow[3];
    SgxCam cam; SgxScales scale, inv_sigma2; float log_scale_factor, th;
    int *best_idx, *best_dist;
};
SGX_KERNEL(256) k_fuse_search(SgxFuseArgs A)
{
    SGX_THREADS_BEGIN(tid)
    const int i = (int)blockIdx.x * 256 + tid;
    if (i < A.nm) {
        int bestDist = 256, bestIdx = -1;
        const float *P = A.m_xw + 3 * (size_t)i;
        bool ok = !A.m_skip[i];
        const float pcx = sgx_gemm3(A.Rcw[0], P, A.tcw[0]), pcy = sgx_gemm3(A.Rcw[1], P, A.tcw[1]), pcz = sgx_gemm3(A.Rcw[2], P, A.tcw[2]);
        ok = ok && !(pcz < 0.0f);
        const float invz = 1 / pcz, x = pcx * invz, y = pcy * invz;
        const float u = A.cam.fx * x + A.cam.cx, v = A.cam.fy * y + A.cam.cy;
        ok = ok && (u >= A.cam.minX && u < A.cam.maxX && v >= A.cam.minY && v < A.cam.maxY);
        const float ur = u - A.cam.bf * invz;
        const float maxDistance = 1.2f * A.m_max_dist[i], minDistance = 0.8f * A.m_min_dist[i];
        const float po0 = P[0] - A.Ow[0], po1 = P[1] - A.Ow[1], po2 = P[2] - A.Ow[2];
        const float dist3D = (float)sqrt((double)po0 * po0 + (double)po1 * po1 + (double)po2 * po2);
        ok = ok && !(dist3D < minDistance || dist3D > maxDistance);
        const double dot = (double)po0 * A.m_normal[3 * (size_t)i] + (double)po1 * A.m_normal[3 * (size_t)i + 1] + (double)po2 * A.m_normal[3 * (size_t)i + 2];
        ok = ok && !(dot < 0.5 * (double)dist3D);
        if (ok) {
            int lvl = (int)ceilf((float)log((double)(A.m_max_dist[i] / dist3D)) / A.log_scale_factor);       // MapPoint::PredictScale(dist, pKF), logf in the reference
            if (lvl < 0) lvl = 0; else if (lvl >= A.nlevels) lvl = A.nlevels - 1;
            const float r = A.th * A.scale.s[lvl];
            const float invW = 64.0f / (A.cam.maxX - A.cam.minX), invH = 48.0f / (A.cam.maxY - A.cam.minY);
            int x0 = (int)floorf((u - A.cam.minX - r) * invW), x1 = (int)ceilf((u - A.cam.minX + r) * invW);
            int y0 = (int)floorf((v - A.cam.minY - r) * invH), y1 = (int)ceilf((v - A.cam.minY + r) * invH);
            x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, 63); y1 = min(y1, 47);
            const uint32_t *dm = A.m_desc + (size_t)i * 8;
            if (x0 < 64 && y0 < 48)
                for (int ix = x0; ix <= x1; ix++) for (int iy = y0; iy <= y1; iy++) {
                    const int c = ix * 48 + iy;
                    for (int q = A.cell_start[c]; q < A.cell_start[c + 1]; q++) {
                        const int idx = A.cell_items[q];
                        const float *kp = (const float *)(A.keys + (size_t)idx * 28);
                        if (!(fabsf(kp[0] - u) < r && fabsf(kp[1] - v) < r)) continue;
                        const int kpLevel = ((const int *)kp)[5];
                        if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
                        const float ex = u - kp[0], ey = v - kp[1];
                        if (A.uright[idx] >= 0) {
                            const float er = ur - A.uright[idx];
                            const float e2 = ex * ex + ey * ey + er * er;
                            if ((double)(e2 * A.inv_sigma2.s[kpLevel]) > 7.8) continue;
                        } else {
                            const float e2 = ex * ex + ey * ey;
                            if ((double)(e2 * A.inv_sigma2.s[kpLevel]) > 5.99) continue;
                        }
                        const int dist = sgx_hamming256(dm, A.desc + (size_t)idx * 8);
                        if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
                    }
                }
        }
        const bool fused = bestDist <= SGX_TH_LOW;
        A.best_idx[i] = fused ? bestIdx : -1; A.best_dist[i] = fused ? bestDist : 256;
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_match_project_kf: ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1474-1601),
// the relocalisation matcher.  The reference walks the keyframe's map points in index order; a point takes the best still-free keypoint of its window ("free" = no map
// point on entry and not taken by an earlier point of this loop).  That greedy order is resolved exactly as in k_match_project_frame: lock[k] = smallest point index that
// took keypoint k; every sweep lets all points choose in parallel against the previous sweep's locks; point i is final after i + 1 sweeps, the loop stops at the first
// sweep that changes no lock.  Candidates are scanned in GetFeaturesInArea order (grid column, grid row, insertion order) with a strict '<', so ties fall as in the reference.
// One workgroup (a single frame per call: relocalisation is a rare, sequential event); CSR grid built on the host.
// ---------------------------------------------------------------------------------------------
struct SgxKfProjArgs {
    int nc, nk, nlevels, orb_dist, check_ori;
    float log_scale_factor, th;
    float Rcw[3][3], tcw[3], Ow[3];
    SgxCam cam; SgxScales scale;
    const uint8_t *ckeys; const uint32_t *cdesc; const uint8_t *c_has_mp; const int *cell_start; const int *cell_items;
    const uint8_t *kf_keys; const uint8_t *kf_ok; const float *m_xw, *m_min_dist, *m_max_dist; const uint32_t *m_desc;
    int *lock_a, *lock_b, *choice, *cur_match, *nmatches;
};

SGX_KERNEL(1024) k_match_project_kf(SgxKfProjArgs A)
{
    SGX_LDS int hist[SGX_HISTO], bad[SGX_HISTO];
    SGX_LDS int s_changed, s_total, s_rejected;
    const int NT = (int)blockDim.x;
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < A.nc; k += NT) { A.lock_a[k] = A.c_has_mp[k] ? -1 : 0x7FFFFFFF; A.cur_match[k] = -1; }
    for (int i = tid; i < SGX_HISTO; i += NT) { hist[i] = 0; bad[i] = 0; }
    if (tid == 0) { s_total = 0; s_rejected = 0; }
    SGX_THREADS_END
    SGX_SYNC();
    int *lock_cur = A.lock_a, *lock_new = A.lock_b;
    const float invW = 64.0f / (A.cam.maxX - A.cam.minX), invH = 48.0f / (A.cam.maxY - A.cam.minY);
    for (int sweep = 0; sweep < A.nk + 2; sweep++) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) s_changed = 0;
        for (int k = tid; k < A.nc; k += NT) lock_new[k] = A.c_has_mp[k] ? -1 : 0x7FFFFFFF;
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < A.nk; i += NT) {
            int best = -1;
            if (A.kf_ok[i]) {
                const float *xw = A.m_xw + 3 * (size_t)i;
                const float xc = sgx_gemm3(A.Rcw[0], xw, A.tcw[0]), yc = sgx_gemm3(A.Rcw[1], xw, A.tcw[1]), zc = sgx_gemm3(A.Rcw[2], xw, A.tcw[2]);
                const float invzc = (float)(1.0 / (double)zc);
                const float u = A.cam.fx * xc * invzc + A.cam.cx, v = A.cam.fy * yc * invzc + A.cam.cy;
                bool ok = !(u < A.cam.minX || u > A.cam.maxX) && !(v < A.cam.minY || v > A.cam.maxY);
                const float po0 = xw[0] - A.Ow[0], po1 = xw[1] - A.Ow[1], po2 = xw[2] - A.Ow[2];
                const float dist3D = (float)sqrt((double)po0 * po0 + (double)po1 * po1 + (double)po2 * po2);
                ok = ok && !(dist3D < 0.8f * A.m_min_dist[i] || dist3D > 1.2f * A.m_max_dist[i]);
                if (ok) {
                    int lvl = (int)ceilf((float)log((double)(A.m_max_dist[i] / dist3D)) / A.log_scale_factor);      // MapPoint::PredictScale(dist, &CurrentFrame)
                    if (lvl < 0) lvl = 0; else if (lvl >= A.nlevels) lvl = A.nlevels - 1;
                    const float r = A.th * A.scale.s[lvl];
                    const int minLevel = lvl - 1, maxLevel = lvl + 1;
                    int x0 = (int)floorf((u - A.cam.minX - r) * invW), x1 = (int)ceilf((u - A.cam.minX + r) * invW);
                    int y0 = (int)floorf((v - A.cam.minY - r) * invH), y1 = (int)ceilf((v - A.cam.minY + r) * invH);
                    x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, 63); y1 = min(y1, 47);
                    const uint32_t *dm = A.m_desc + (size_t)i * 8;
                    int bestDist = 256;
                    if (x0 < 64 && y0 < 48)
                        for (int ix = x0; ix <= x1; ix++) for (int iy = y0; iy <= y1; iy++) {
                            const int c = ix * 48 + iy;
                            for (int q = A.cell_start[c]; q < A.cell_start[c + 1]; q++) {
                                const int k = A.cell_items[q];
                                const float *kp = (const float *)(A.ckeys + (size_t)k * 28);
                                const int oct = ((const int *)kp)[5];
                                if (oct < minLevel) continue;                       // bCheckLevels: minLevel > 0 || maxLevel >= 0 always holds here (maxLevel >= 1)
                                if (oct > maxLevel) continue;
                                if (!(fabsf(kp[0] - u) < r && fabsf(kp[1] - v) < r)) continue;
                                if (lock_cur[k] < i) continue;                     // holds a map point: on entry (-1) or taken by an earlier point of this loop
                                const int dist = sgx_hamming256(dm, A.cdesc + (size_t)k * 8);
                                if (dist < bestDist) { bestDist = dist; best = k; }
                            }
                        }
                    if (bestDist > A.orb_dist) best = -1;
                }
            }
            A.choice[i] = best;
            if (best >= 0) sgx_atomic_min_i32(&lock_new[best], i);
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int k = tid; k < A.nc; k += NT) if (lock_new[k] != lock_cur[k]) s_changed = 1;
        SGX_THREADS_END
        SGX_SYNC();
        int *t = lock_cur; lock_cur = lock_new; lock_new = t;
        if (!s_changed) break;
    }
    // assignments (at the fixpoint a keypoint is chosen by exactly one point: the one that holds its lock) and the rotation histogram
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < A.nk; i += NT) {
        const int k = A.choice[i];
        if (k < 0) continue;
        A.cur_match[k] = i;
        sgx_atomic_add(&s_total, 1);
        if (A.check_ori) {
            const float ka = ((const float *)(A.kf_keys + (size_t)i * 28))[3], ca = ((const float *)(A.ckeys + (size_t)k * 28))[3];
            float rot = ka - ca;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
            if (bin == SGX_HISTO) bin = 0;
            sgx_atomic_add(&hist[bin], 1);
        }
    }
    SGX_THREADS_END
    SGX_SYNC();
    if (A.check_ori) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {                                         // ComputeThreeMaxima, ORBmatcher.cc:1603-1644
            int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < SGX_HISTO; i++) {
                const int s = hist[i];
                if (s > m1) { m3 = m2; m2 = m1; m1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > m2) { m3 = m2; m2 = s; i3 = i2; i2 = i; }
                else if (s > m3) { m3 = s; i3 = i; }
            }
            if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
            else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
            for (int i = 0; i < SGX_HISTO; i++) bad[i] = (i != i1 && i != i2 && i != i3);
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < A.nk; i += NT) {
            const int k = A.choice[i];
            if (k < 0) continue;
            const float ka = ((const float *)(A.kf_keys + (size_t)i * 28))[3], ca = ((const float *)(A.ckeys + (size_t)k * 28))[3];
            float rot = ka - ca;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
            if (bin == SGX_HISTO) bin = 0;
            if (bad[bin]) { A.cur_match[k] = -1; sgx_atomic_add(&s_rejected, 1); }
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) *A.nmatches = s_total - s_rejected;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// Loop-closing matchers that project map points through a similarity into a keyframe (LoopClosing::ComputeSim3 / SearchAndFuse):
//   ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)            ORBmatcher.cc:981-1101   k_sim3_search, SGX_S3_NORMAL
//   ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) ORBmatcher.cc:1106-1330  k_sim3_search twice (SGX_S3_TWO_STEP | SGX_S3_CAM_DIST), then the agreement check
//   ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)    ORBmatcher.cc:292-407    k_sim3_search_locked, SGX_S3_NORMAL | SGX_S3_FLOAT_INVZ
// All three share one body: transform, depth / image / distance (/ viewing-angle) gates, MapPoint::PredictScale, KeyFrame::GetFeaturesInArea in the reference's scan order
// (KeyFrame.cc:570-609) with the octave gate [level - 1, level], best Hamming distance with first-wins ties.
// ---------------------------------------------------------------------------------------------
#define SGX_S3_TWO_STEP   1     // p = R2 * (R1 * xw + t1) + t2   (SearchBySim3: world -> source camera -> target camera)
#define SGX_S3_CAM_DIST   2     // dist3D = |p|  (SearchBySim3) instead of |xw - Ow|
#define SGX_S3_NORMAL     4     // viewing-angle gate PO . Pn >= 0.5 dist
#define SGX_S3_FLOAT_INVZ 8     // invz = 1 / z in float (:333); otherwise (float)(1.0 / z) (:1022, :1170)
struct SgxSim3ProjArgs {
    int nk, nm, nlevels, flags, th_accept;
    const uint8_t *keys; const uint32_t *desc;          // target keyframe: mvKeysUn, mDescriptors
    const int *cell_start, *cell_items;                 // CSR of its grid: cell (ix, iy) -> ix * 48 + iy
    const float *m_xw, *m_normal, *m_min_dist, *m_max_dist; const uint32_t *m_desc; const uint8_t *m_skip;
    float R1[3][3], t1[3], R2[3][3], t2[3], Ow[3];
    SgxCam cam; SgxScales scale; float log_scale_factor, th;
    const uint8_t *taken_in;                            // locked variant: vpMatched[k] != NULL on entry
    int *lock_a, *lock_b;
    int *best_idx, *best_dist;                          // per candidate
    int *matched_out, *nmatches;                        // locked variant: per keypoint, total
};

// best keypoint for candidate i (-1: none); lock (may be NULL): keypoints with lock[k] < i are held by an earlier candidate (or on entry: -1)
SGX_DEV int sgx_sim3_project_one(const SgxSim3ProjArgs &A, int i, const int *lock, int *dist_out)
{
    int bestDist = 0x7FFFFFFF, bestIdx = -1;
    const float *P = A.m_xw + 3 * (size_t)i;
    bool ok = !A.m_skip[i];
    float pc[3] = { sgx_gemm3(A.R1[0], P, A.t1[0]), sgx_gemm3(A.R1[1], P, A.t1[1]), sgx_gemm3(A.R1[2], P, A.t1[2]) };
    if (A.flags & SGX_S3_TWO_STEP) {
        const float q[3] = { sgx_gemm3(A.R2[0], pc, A.t2[0]), sgx_gemm3(A.R2[1], pc, A.t2[1]), sgx_gemm3(A.R2[2], pc, A.t2[2]) };
        pc[0] = q[0]; pc[1] = q[1]; pc[2] = q[2];
    }
    ok = ok && !(pc[2] < 0.0f);
    const float invz = (A.flags & SGX_S3_FLOAT_INVZ) ? 1 / pc[2] : (float)(1.0 / (double)pc[2]);
    const float x = pc[0] * invz, y = pc[1] * invz;
    const float u = A.cam.fx * x + A.cam.cx, v = A.cam.fy * y + A.cam.cy;
    ok = ok && (u >= A.cam.minX && u < A.cam.maxX && v >= A.cam.minY && v < A.cam.maxY);                    // KeyFrame::IsInImage
    const float maxDistance = 1.2f * A.m_max_dist[i], minDistance = 0.8f * A.m_min_dist[i];
    float d0, d1, d2;
    if (A.flags & SGX_S3_CAM_DIST) { d0 = pc[0]; d1 = pc[1]; d2 = pc[2]; } else { d0 = P[0] - A.Ow[0]; d1 = P[1] - A.Ow[1]; d2 = P[2] - A.Ow[2]; }
    const float dist3D = (float)sqrt((double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2);
    ok = ok && !(dist3D < minDistance || dist3D > maxDistance);
    if (A.flags & SGX_S3_NORMAL) {
        const double dot = (double)d0 * A.m_normal[3 * (size_t)i] + (double)d1 * A.m_normal[3 * (size_t)i + 1] + (double)d2 * A.m_normal[3 * (size_t)i + 2];
        ok = ok && !(dot < 0.5 * (double)dist3D);
    }
    if (ok) {
        int lvl = (int)ceilf((float)log((double)(A.m_max_dist[i] / dist3D)) / A.log_scale_factor);           // MapPoint::PredictScale(dist, pKF), logf in the reference
        if (lvl < 0) lvl = 0; else if (lvl >= A.nlevels) lvl = A.nlevels - 1;
        const float r = A.th * A.scale.s[lvl];
        const float invW = 64.0f / (A.cam.maxX - A.cam.minX), invH = 48.0f / (A.cam.maxY - A.cam.minY);
        int x0 = (int)floorf((u - A.cam.minX - r) * invW), x1 = (int)ceilf((u - A.cam.minX + r) * invW);
        int y0 = (int)floorf((v - A.cam.minY - r) * invH), y1 = (int)ceilf((v - A.cam.minY + r) * invH);
        x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, 63); y1 = min(y1, 47);
        const uint32_t *dm = A.m_desc + (size_t)i * 8;
        if (x0 < 64 && y0 < 48)
            for (int ix = x0; ix <= x1; ix++) for (int iy = y0; iy <= y1; iy++) {
                const int c = ix * 48 + iy;
                for (int q = A.cell_start[c]; q < A.cell_start[c + 1]; q++) {
                    const int idx = A.cell_items[q];
                    const float *kp = (const float *)(A.keys + (size_t)idx * 28);
                    if (!(fabsf(kp[0] - u) < r && fabsf(kp[1] - v) < r)) continue;
                    if (lock && lock[idx] < i) continue;
                    const int kpLevel = ((const int *)kp)[5];
                    if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
                    const int dist = sgx_hamming256(dm, A.desc + (size_t)idx * 8);
                    if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
                }
            }
    }
    if (bestDist > A.th_accept) bestIdx = -1;
    *dist_out = bestIdx >= 0 ? bestDist : 256;
    return bestIdx;
}

SGX_KERNEL(256) k_sim3_search(SgxSim3ProjArgs A)
{
    SGX_THREADS_BEGIN(tid)
    const int i = (int)blockIdx.x * 256 + tid;
    if (i < A.nm) {
        int d;
        const int b = sgx_sim3_project_one(A, i, nullptr, &d);
        A.best_idx[i] = b; A.best_dist[i] = d;
    }
    SGX_THREADS_END
}

// The reference walks vpPoints in order and a point skips keypoints whose vpMatched slot is already filled (:381), by the caller or by an earlier point of the loop.
// Resolved exactly as in k_match_project_kf: lock[k] = smallest candidate index that took keypoint k (-1: filled on entry); sweeps until no lock changes.
SGX_KERNEL(1024) k_sim3_search_locked(SgxSim3ProjArgs A)
{
    SGX_LDS int s_changed, s_total;
    const int NT = (int)blockDim.x;
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < A.nk; k += NT) { A.lock_a[k] = A.taken_in[k] ? -1 : 0x7FFFFFFF; A.matched_out[k] = -1; }
    if (tid == 0) s_total = 0;
    SGX_THREADS_END
    SGX_SYNC();
    int *lock_cur = A.lock_a, *lock_new = A.lock_b;
    for (int sweep = 0; sweep < A.nm + 2; sweep++) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) s_changed = 0;
        for (int k = tid; k < A.nk; k += NT) lock_new[k] = A.taken_in[k] ? -1 : 0x7FFFFFFF;
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < A.nm; i += NT) {
            int d;
            const int b = sgx_sim3_project_one(A, i, lock_cur, &d);
            A.best_idx[i] = b; A.best_dist[i] = d;
            if (b >= 0) sgx_atomic_min_i32(&lock_new[b], i);
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int k = tid; k < A.nk; k += NT) if (lock_new[k] != lock_cur[k]) s_changed = 1;
        SGX_THREADS_END
        SGX_SYNC();
        int *t = lock_cur; lock_cur = lock_new; lock_new = t;
        if (!s_changed) break;
    }
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < A.nm; i += NT) {
        const int k = A.best_idx[i];
        if (k < 0) continue;
        A.matched_out[k] = i;                       // at the fixpoint a keypoint is chosen by exactly one candidate
        sgx_atomic_add(&s_total, 1);
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) *A.nmatches = s_total;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_search_initialization: ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:407-522), the monocular initialiser's
// matcher.  The reference walks F1's level-0 keypoints in order; a keypoint may take a keypoint of F2 from an earlier one when its distance is strictly smaller
// (vMatchedDistance), so the outer loop is inherently sequential.  One workgroup: the outer loop runs in order, the window search of a keypoint is spread over the
// threads (the cells of one grid column are contiguous in the CSR grid): pass A finds (smallest distance, first in GetFeaturesInArea order) with one LDS atomic min on
// distance << 22 | ordinal, pass B the second smallest distance among the others; thread 0 applies the accept / steal rules.  A rare, one-frame-pair event: latency over throughput.
// ---------------------------------------------------------------------------------------------
struct SgxInitSearchArgs {
    int n1, n2, window, check_ori; float nnratio;
    SgxCam cam;
    const uint8_t *keys1, *keys2; const uint32_t *desc1, *desc2;
    const int *cell_start, *cell_items;                 // CSR grid of F2: cell (ix, iy) -> ix * 48 + iy
    float *prev_matched;                                // in / out
    int *matches12, *matches21, *dist21, *nmatches;
};

SGX_KERNEL(256) k_search_initialization(SgxInitSearchArgs A)
{
    SGX_LDS int hist[SGX_HISTO], bad[SGX_HISTO];
    SGX_LDS int s_key, s_second, s_best_idx, s_total;
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < A.n1; i += 256) A.matches12[i] = -1;
    for (int i = tid; i < A.n2; i += 256) { A.matches21[i] = -1; A.dist21[i] = 0x7FFFFFFF; }
    for (int i = tid; i < SGX_HISTO; i += 256) { hist[i] = 0; bad[i] = 0; }
    if (tid == 0) s_total = 0;
    SGX_THREADS_END
    SGX_SYNC();
    const float invW = 64.0f / (A.cam.maxX - A.cam.minX), invH = 48.0f / (A.cam.maxY - A.cam.minY), r = (float)A.window;
    for (int i1 = 0; i1 < A.n1; i1++) {
        const float *kp1 = (const float *)(A.keys1 + (size_t)i1 * 28);
        if (((const int *)kp1)[5] > 0) continue;                                   // level1 > 0 (:424-426); uniform
        const float x = A.prev_matched[2 * i1], y = A.prev_matched[2 * i1 + 1];
        int x0 = (int)floorf((x - A.cam.minX - r) * invW), x1 = (int)ceilf((x - A.cam.minX + r) * invW);
        int y0 = (int)floorf((y - A.cam.minY - r) * invH), y1 = (int)ceilf((y - A.cam.minY + r) * invH);
        x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, 63); y1 = min(y1, 47);
        if (x0 >= 64 || y0 >= 48 || x1 < 0 || y1 < 0) continue;                    // Frame::GetFeaturesInArea early returns
        const uint32_t *da = A.desc1 + (size_t)i1 * 8;
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) { s_key = 0x7FFFFFFF; s_second = 0x7FFFFFFF; s_best_idx = -1; }
        SGX_THREADS_END
        SGX_SYNC();
        for (int pass = 0; pass < 2; pass++) {
            SGX_THREADS_BEGIN(tid)
            int ordinal = 0;
            for (int ix = x0; ix <= x1; ix++) {
                const int s = A.cell_start[ix * 48 + y0], e = A.cell_start[ix * 48 + y1 + 1];
                for (int q = s + tid; q < e; q += 256) {
                    const int i2 = A.cell_items[q];
                    const float *kp2 = (const float *)(A.keys2 + (size_t)i2 * 28);
                    if (((const int *)kp2)[5] != 0) continue;                                       // minLevel = maxLevel = level1 = 0
                    if (!(fabsf(kp2[0] - x) < r && fabsf(kp2[1] - y) < r)) continue;
                    const int dist = sgx_hamming256(da, A.desc2 + (size_t)i2 * 8);
                    if (A.dist21[i2] <= dist) continue;                                             // vMatchedDistance (:444-445)
                    const int key = (dist << 22) | (ordinal + (q - s));
                    if (pass == 0) sgx_atomic_min_i32(&s_key, key);
                    else if (key == s_key) s_best_idx = i2;
                    else sgx_atomic_min_i32(&s_second, dist);
                }
                ordinal += e - s;
            }
            SGX_THREADS_END
            SGX_SYNC();
        }
        SGX_THREADS_BEGIN(tid)
        if (tid == 0 && s_best_idx >= 0) {
            const int bestDist = s_key >> 22, bestDist2 = s_second, bestIdx2 = s_best_idx;
            if (bestDist <= SGX_TH_LOW && (float)bestDist < (float)bestDist2 * A.nnratio) {
                if (A.matches21[bestIdx2] >= 0) { A.matches12[A.matches21[bestIdx2]] = -1; s_total--; }
                A.matches12[i1] = bestIdx2; A.matches21[bestIdx2] = i1; A.dist21[bestIdx2] = bestDist; s_total++;
                if (A.check_ori) {
                    float rot = kp1[3] - ((const float *)(A.keys2 + (size_t)bestIdx2 * 28))[3];
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
                    if (bin == SGX_HISTO) bin = 0;
                    hist[bin]++;                                                                    // stays counted when the match is stolen later, as rotHist does
                }
            }
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    if (A.check_ori) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {
            int m1 = 0, m2 = 0, m3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < SGX_HISTO; i++) {
                const int s = hist[i];
                if (s > m1) { m3 = m2; m2 = m1; m1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > m2) { m3 = m2; m2 = s; i3 = i2; i2 = i; }
                else if (s > m3) { m3 = s; i3 = i; }
            }
            if ((float)m2 < 0.1f * (float)m1) { i2 = -1; i3 = -1; }
            else if ((float)m3 < 0.1f * (float)m1) { i3 = -1; }
            for (int i = 0; i < SGX_HISTO; i++) bad[i] = (i != i1 && i != i2 && i != i3);
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < A.n1; i += 256) {
            const int j = A.matches12[i];
            if (j < 0) continue;
            float rot = ((const float *)(A.keys1 + (size_t)i * 28))[3] - ((const float *)(A.keys2 + (size_t)j * 28))[3];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
            if (bin == SGX_HISTO) bin = 0;
            if (bad[bin]) { A.matches12[i] = -1; sgx_atomic_add(&s_total, -1); }
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < A.n1; i += 256) {                                         // update vbPrevMatched (:516-519)
        const int j = A.matches12[i];
        if (j >= 0) { const float *kp2 = (const float *)(A.keys2 + (size_t)j * 28); A.prev_matched[2 * i] = kp2[0]; A.prev_matched[2 * i + 1] = kp2[1]; }
    }
    if (tid == 0) *A.nmatches = s_total;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// MapPoint post-steps of the optimisers and of map-point creation, batched over points:
//   k_mappoint_normal_depth   MapPoint::UpdateNormalAndDepth  (src/sg-slam/src/MapPoint.cc:330-371; Optimizer.cc:227,776,1042, LocalMapping.cc:152,444,527, Tracking.cc:1234)
//   k_mappoint_distinctive    MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:242-307)
// Observations arrive as CSR in the order the reference walks mObservations (its float sums are order dependent).
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_mappoint_normal_depth(int n, const float *xw, const int *obs_start, const float *obs_center, const float *ref_center, const int *ref_level,
                                        SgxScales scale, int nlevels, float *normal, float *min_dist, float *max_dist)
{
    SGX_THREADS_BEGIN(tid)
    const int p = (int)blockIdx.x * 256 + tid;
    if (p < n) {
        const int s = obs_start[p], e = obs_start[p + 1];
        if (e > s) {
            const float *P = xw + 3 * (size_t)p;
            float n0 = 0.f, n1 = 0.f, n2 = 0.f;
            for (int q = s; q < e; q++) {
                const float d0 = P[0] - obs_center[3 * (size_t)q], d1 = P[1] - obs_center[3 * (size_t)q + 1], d2 = P[2] - obs_center[3 * (size_t)q + 2];
                const double len = sqrt((double)d0 * d0 + (double)d1 * d1 + (double)d2 * d2);
                const float inv = (float)(1.0 / len);
                n0 = n0 + d0 * inv; n1 = n1 + d1 * inv; n2 = n2 + d2 * inv;
            }
            const float c0 = P[0] - ref_center[3 * (size_t)p], c1 = P[1] - ref_center[3 * (size_t)p + 1], c2 = P[2] - ref_center[3 * (size_t)p + 2];
            const float dist = (float)sqrt((double)c0 * c0 + (double)c1 * c1 + (double)c2 * c2);
            const float mx = dist * scale.s[ref_level[p]];
            max_dist[p] = mx; min_dist[p] = mx / scale.s[nlevels - 1];
            const float invn = (float)(1.0 / (double)(e - s));
            normal[3 * (size_t)p] = n0 * invn; normal[3 * (size_t)p + 1] = n1 * invn; normal[3 * (size_t)p + 2] = n2 * invn;
        }
    }
    SGX_THREADS_END
}

// One workgroup (64 threads) per point; thread i owns row i of the N x N distance matrix (rows beyond 64 in further rounds).  The median of a row = element
// (int)(0.5 * (N - 1)) of the sorted row = the smallest v with count(d <= v) > k: nine bisection steps over the 257 possible distances, distances recomputed
// (no N x N storage).  The row with the least median wins, the first one on ties (strict '<' in row order): LDS atomic min on median << 16 | row.
SGX_KERNEL(64) k_mappoint_distinctive(int n, const int *obs_start, const uint32_t *obs_desc, int *best)
{
    SGX_LDS int s_key;
    const int p = (int)blockIdx.x;
    const int s = obs_start[p], N = obs_start[p + 1] - s;
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) s_key = 0x7FFFFFFF;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int k = (int)(0.5 * (double)(N - 1));
    for (int i = tid; i < N; i += 64) {
        uint32_t di[8];
#pragma unroll
        for (int w = 0; w < 8; w++) di[w] = obs_desc[(size_t)(s + i) * 8 + w];
        int lo = 0, hi = 256;                                          // smallest v in [0, 256] with count(d <= v) >= k + 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; j++) cnt += (i == j ? 0 : sgx_hamming256(di, obs_desc + (size_t)(s + j) * 8)) <= mid ? 1 : 0;
            if (cnt >= k + 1) hi = mid; else lo = mid + 1;
        }
        sgx_atomic_min_i32(&s_key, (lo << 16) | i);
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) best[p] = N > 0 ? (s_key & 0xFFFF) : -1;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_triangulate_pairs: the per-pair body of LocalMapping::CreateNewMapPoints (src/sg-slam/src/LocalMapping.cc:283-421) for the pairs SearchForTriangulation returned:
// parallax test, linear triangulation (cv::SVD::compute on the 4 x 4 system = OpenCV's one-sided Jacobi, JacobiSVDImpl_<float>) or the stereo unprojection of the better
// side, positive depth in both keyframes, the chi-square reprojection gates (mono 5.991 / stereo 7.8), the scale-consistency gate.  Pairs are independent: one thread each.
// The map mutations that follow (:402-419) stay with the caller.
// ---------------------------------------------------------------------------------------------
struct SgxNewPointArgs {
    int npairs;
    const int *pairs;
    const uint8_t *keys1_un, *keys1, *keys2_un, *keys2; const float *ur1, *dp1, *ur2, *dp2;
    float Tcw1[16], Tcw2[16], Ow1[3], Ow2[3];
    float fx, fy, cx, cy, mbf, ratio_factor;
    SgxScales scale, sigma2;
    uint8_t *ok; float *x3d;
};

SGX_DEV void sgx_jacobi_svd4_vt3(const float *A, float *v)
{
    const int n = 4, m = 4; const float eps = FLT_EPSILON * 2;
    float At[16], Vt[16]; double W[4];
    for (int i = 0; i < 4; i++) for (int k = 0; k < 4; k++) At[4 * i + k] = A[4 * k + i];
    for (int i = 0; i < n; i++) { double sd = 0; for (int k = 0; k < m; k++) { const float t = At[4 * i + k]; sd += (double)t * t; } W[i] = sd; for (int k = 0; k < n; k++) Vt[4 * i + k] = i == k ? 1.f : 0.f; }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                float *Ai = At + 4 * i, *Aj = At + 4 * j;
                double a = W[i], p = 0, b = W[j];
                for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
                if (fabs(p) <= (double)eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot(p, beta);
                float c, s;
                if (beta < 0) { const double delta = (gamma - beta) * 0.5; s = (float)sqrt(delta / gamma); c = (float)(p / (gamma * (double)s * 2)); }
                else { c = (float)sqrt((gamma + beta) / (gamma * 2)); s = (float)(p / (gamma * (double)c * 2)); }
                a = b = 0;
                for (int k = 0; k < m; k++) { const float t0 = c * Ai[k] + s * Aj[k], t1 = -s * Ai[k] + c * Aj[k]; Ai[k] = t0; Aj[k] = t1; a += (double)t0 * t0; b += (double)t1 * t1; }
                W[i] = a; W[j] = b; changed = true;
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
    for (int k = 0; k < 4; k++) v[k] = Vt[12 + k];
}

SGX_DEV double sgx_dot3d(const float *a, const float *b) { double r = 0; for (int i = 0; i < 3; i++) r += (double)a[i] * (double)b[i]; return r; }
SGX_DEV float sgx_gemm3_t(const float *T, int col, const float *b, float c, double beta)      // row `col` of the TRANSPOSE of the 3 x 3 block of T (4-float rows) times b, + beta * c
{
    const float t = T[col] * b[0] + T[4 + col] * b[1] + T[8 + col] * b[2];
    return (float)((double)t * 1.0 + beta * (double)c);
}

SGX_KERNEL(256) k_triangulate_pairs(SgxNewPointArgs A)
{
    SGX_THREADS_BEGIN(tid)
    const int q = (int)blockIdx.x * 256 + tid;
    if (q < A.npairs) {
        uint8_t okq = 0; float X[3] = { 0.f, 0.f, 0.f };
        do {
            const int idx1 = A.pairs[2 * q], idx2 = A.pairs[2 * q + 1];
            const float *kp1 = (const float *)(A.keys1_un + (size_t)idx1 * 28), *kp2 = (const float *)(A.keys2_un + (size_t)idx2 * 28);
            const int oct1 = ((const int *)kp1)[5], oct2 = ((const int *)kp2)[5];
            const float invfx = 1.0f / A.fx, invfy = 1.0f / A.fy, mb = A.mbf / A.fx;
            const float kp1_ur = A.ur1[idx1], kp2_ur = A.ur2[idx2];
            const bool bStereo1 = kp1_ur >= 0, bStereo2 = kp2_ur >= 0;
            const float xn1[3] = { (kp1[0] - A.cx) * invfx, (kp1[1] - A.cy) * invfy, 1.0f }, xn2[3] = { (kp2[0] - A.cx) * invfx, (kp2[1] - A.cy) * invfy, 1.0f };
            const float ray1[3] = { sgx_gemm3_t(A.Tcw1, 0, xn1, 0.f, 0.0), sgx_gemm3_t(A.Tcw1, 1, xn1, 0.f, 0.0), sgx_gemm3_t(A.Tcw1, 2, xn1, 0.f, 0.0) };
            const float ray2[3] = { sgx_gemm3_t(A.Tcw2, 0, xn2, 0.f, 0.0), sgx_gemm3_t(A.Tcw2, 1, xn2, 0.f, 0.0), sgx_gemm3_t(A.Tcw2, 2, xn2, 0.f, 0.0) };
            const float cosParallaxRays = (float)(sgx_dot3d(ray1, ray2) / (sqrt(sgx_dot3d(ray1, ray1)) * sqrt(sgx_dot3d(ray2, ray2))));
            float cosParallaxStereo = cosParallaxRays + 1, cs1 = cosParallaxStereo, cs2 = cosParallaxStereo;
            if (bStereo1) cs1 = cosf(2 * atan2f(mb / 2, A.dp1[idx1]));
            else if (bStereo2) cs2 = cosf(2 * atan2f(mb / 2, A.dp2[idx2]));
            cosParallaxStereo = cs1 < cs2 ? cs1 : cs2;
            if (cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && (bStereo1 || bStereo2 || (double)cosParallaxRays < 0.9998)) {
                float M[16], v[4];
                for (int k = 0; k < 4; k++) {
                    M[k] = xn1[0] * A.Tcw1[8 + k] - A.Tcw1[k]; M[4 + k] = xn1[1] * A.Tcw1[8 + k] - A.Tcw1[4 + k];
                    M[8 + k] = xn2[0] * A.Tcw2[8 + k] - A.Tcw2[k]; M[12 + k] = xn2[1] * A.Tcw2[8 + k] - A.Tcw2[4 + k];
                }
                sgx_jacobi_svd4_vt3(M, v);
                if (v[3] == 0) break;
                const float inv = (float)(1.0 / (double)v[3]);
                X[0] = v[0] * inv; X[1] = v[1] * inv; X[2] = v[2] * inv;
            } else if (bStereo1 && cs1 < cs2) {
                const float z = A.dp1[idx1];
                if (!(z > 0)) break;
                const float *kd = (const float *)(A.keys1 + (size_t)idx1 * 28);
                const float xc[3] = { (kd[0] - A.cx) * z * invfx, (kd[1] - A.cy) * z * invfy, z };
                for (int i = 0; i < 3; i++) X[i] = sgx_gemm3_t(A.Tcw1, i, xc, A.Ow1[i], 1.0);
            } else if (bStereo2 && cs2 < cs1) {
                const float z = A.dp2[idx2];
                if (!(z > 0)) break;
                const float *kd = (const float *)(A.keys2 + (size_t)idx2 * 28);
                const float xc[3] = { (kd[0] - A.cx) * z * invfx, (kd[1] - A.cy) * z * invfy, z };
                for (int i = 0; i < 3; i++) X[i] = sgx_gemm3_t(A.Tcw2, i, xc, A.Ow2[i], 1.0);
            } else break;
            const float z1 = (float)(sgx_dot3d(A.Tcw1 + 8, X) + (double)A.Tcw1[11]);
            if (z1 <= 0) break;
            const float z2 = (float)(sgx_dot3d(A.Tcw2 + 8, X) + (double)A.Tcw2[11]);
            if (z2 <= 0) break;
            const float s1 = A.sigma2.s[oct1];
            const float x1 = (float)(sgx_dot3d(A.Tcw1, X) + (double)A.Tcw1[3]), y1 = (float)(sgx_dot3d(A.Tcw1 + 4, X) + (double)A.Tcw1[7]);
            const float invz1 = (float)(1.0 / (double)z1);
            {
                const float u1 = A.fx * x1 * invz1 + A.cx, v1 = A.fy * y1 * invz1 + A.cy, eX = u1 - kp1[0], eY = v1 - kp1[1];
                if (!bStereo1) { if ((double)(eX * eX + eY * eY) > 5.991 * (double)s1) break; }
                else { const float u1r = u1 - A.mbf * invz1, eR = u1r - kp1_ur; if ((double)(eX * eX + eY * eY + eR * eR) > 7.8 * (double)s1) break; }
            }
            const float s2 = A.sigma2.s[oct2];
            const float x2 = (float)(sgx_dot3d(A.Tcw2, X) + (double)A.Tcw2[3]), y2 = (float)(sgx_dot3d(A.Tcw2 + 4, X) + (double)A.Tcw2[7]);
            const float invz2 = (float)(1.0 / (double)z2);
            {
                const float u2 = A.fx * x2 * invz2 + A.cx, v2 = A.fy * y2 * invz2 + A.cy, eX = u2 - kp2[0], eY = v2 - kp2[1];
                if (!bStereo2) { if ((double)(eX * eX + eY * eY) > 5.991 * (double)s2) break; }
                else { const float u2r = u2 - A.mbf * invz2, eR = u2r - kp2_ur; if ((double)(eX * eX + eY * eY + eR * eR) > 7.8 * (double)s2) break; }
            }
            const float n1[3] = { X[0] - A.Ow1[0], X[1] - A.Ow1[1], X[2] - A.Ow1[2] }, n2[3] = { X[0] - A.Ow2[0], X[1] - A.Ow2[1], X[2] - A.Ow2[2] };
            const float dist1 = (float)sqrt(sgx_dot3d(n1, n1)), dist2 = (float)sqrt(sgx_dot3d(n2, n2));
            if (dist1 == 0 || dist2 == 0) break;
            const float ratioDist = dist2 / dist1, ratioOctave = A.scale.s[oct1] / A.scale.s[oct2];
            if (ratioDist * A.ratio_factor < ratioOctave || ratioDist > ratioOctave * A.ratio_factor) break;
            okq = 1;
        } while (0);
        A.ok[q] = okq;
        A.x3d[3 * (size_t)q] = okq ? X[0] : 0.f; A.x3d[3 * (size_t)q + 1] = okq ? X[1] : 0.f; A.x3d[3 * (size_t)q + 2] = okq ? X[2] : 0.f;
    }
    SGX_THREADS_END
}
