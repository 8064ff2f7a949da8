// sgx_match_kernels.h — HIP kernels of the ORB matcher + the per-frame glue the matcher needs
// (stereo-from-RGBD, unprojection).  Phase style, see sgx_rt.h.
// Reference behaviour: src/sg-slam/src/ORBmatcher.cc, src/sg-slam/src/Frame.cc (cited per kernel).
#pragma once
#include "sgx_rt.h"

#define SGX_GRID_COLS 64          /* FRAME_GRID_COLS, Frame.h:40 */
#define SGX_GRID_ROWS 48          /* FRAME_GRID_ROWS, Frame.h:39 */
#define SGX_MATCH_CAP 1280        /* max keypoints per frame handled in LDS */
#define SGX_MATCH_THREADS 1024

#include "sgx_match_common.h"


// ---------------------------------------------------------------------------------------------
// k_match_project_frame: ORBmatcher::SearchByProjection(Frame &Cur, const Frame &Last, th, bMono)
// (ORBmatcher.cc:1332-1472), one 1024-thread workgroup per frame pair, thread i <-> last-frame
// map point i.
//
// The reference walks the map points in index order; a current keypoint already holding a map
// point with Observations()>0 is skipped by later map points (:1407-1409), otherwise it may be
// re-assigned (the later map point wins).  Parallel restatement:
//   * candidate set of map point i = exactly the keypoints Frame::GetFeaturesInArea (Frame.cc:354-407)
//     would return: valid grid cell (PosInGrid uses round(), :411-418) inside the floor/ceil cell
//     window, level gate, |dx|<r and |dy|<r; the 64x48 grid is rebuilt in LDS as a CSR index (its
//     per-cell order is irrelevant: the reference's scan order is encoded in the preference key).
//   * preference = min over unlocked candidates of (distance, cell x, cell y, keypoint index) ==
//     "first strictly smaller distance in scan order" (:1423).
//   * locks: lock[k] = smallest i with Observations()>0 whose choice is k; k is unavailable to i'
//     iff lock[k] < i'.  The sequential greedy result is the unique fixpoint of
//     choice_i = best unlocked under the locks of {choice_j, j<i}; Jacobi iteration reaches it
//     (map point i is exact after i+1 sweeps; in practice 1-3 sweeps) and stops when the lock
//     table no longer changes.
//   * rotation histogram (:1436-1469): bins from all assignments (overwritten ones included, as in
//     the reference), three maxima, keypoints that received ANY assignment from a rejected bin are
//     cleared, nmatches = assignments - rejected assignments.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(SGX_MATCH_THREADS) k_match_project_frame(
    int cap, const uint8_t *ckeys_raw, const uint8_t *cdesc, const float *curight, const int *cn, const float *cTcw,
    const uint8_t *lkeys_raw, const int *ln, const uint8_t *l_has_mp, const uint8_t *l_outlier, const float *l_xw,
    const int *l_obs, const uint8_t *l_mpdesc, const float *lTcw,
    SgxCam cam, SgxScales sc, float th, int bMono, int check_ori, int *cur_match, int *nmatches_out)
{
    SGX_LDS float kx[SGX_MATCH_CAP], ky[SGX_MATCH_CAP], kur[SGX_MATCH_CAP], kang[SGX_MATCH_CAP];
    SGX_LDS uint32_t kinfo[SGX_MATCH_CAP];           // octave | cellx<<8 | celly<<16 | valid<<31
    SGX_LDS uint32_t kdesc[SGX_MATCH_CAP * 8];
    SGX_LDS int lock_a[SGX_MATCH_CAP], lock_b[SGX_MATCH_CAP];
    SGX_LDS int choice[SGX_MATCH_CAP];
    SGX_LDS int owner[SGX_MATCH_CAP];
    SGX_LDS int hist[SGX_HISTO], bad[SGX_HISTO];
    SGX_LDS int cell_start[SGX_GRID_COLS * SGX_GRID_ROWS + 1];   // Frame::mGrid as CSR: cell = ix*48 + iy (AssignFeaturesToGrid, Frame.cc:257-272)
    SGX_LDS int cell_fill[SGX_GRID_COLS * SGX_GRID_ROWS];
    SGX_LDS uint16_t cell_list[SGX_MATCH_CAP];
    SGX_LDS int s_changed, s_total, s_rejected, s_ngrid;

    const int f = (int)blockIdx.x;
    SGX_WAVE_PRIORITY(3);          // latency-critical one-workgroup-per-frame kernel (see k_pose_opt)
    const int Nc = min(cn[f], cap), Nl = min(ln[f], cap);
    const int NT = (int)blockDim.x;
    const float *Tc = cTcw + 16 * f, *Tl = lTcw + 16 * f;
    const float invW = (float)SGX_GRID_COLS / (cam.maxX - cam.minX), invH = (float)SGX_GRID_ROWS / (cam.maxY - cam.minY);

    // ---- stage the current frame
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < Nc; k += NT) {
        const float *kp = (const float *)(ckeys_raw + ((size_t)f * cap + k) * 28);
        const float x = kp[0], y = kp[1];
        kx[k] = x; ky[k] = y; kang[k] = kp[3]; kur[k] = curight[(size_t)f * cap + k];
        const int oct = ((const int *)kp)[5];
        const int px = (int)round((double)((x - cam.minX) * invW)), py = (int)round((double)((y - cam.minY) * invH));
        const bool valid = !(px < 0 || px >= SGX_GRID_COLS || py < 0 || py >= SGX_GRID_ROWS);
        kinfo[k] = (uint32_t)(oct & 0xFF) | ((uint32_t)(px & 0xFF) << 8) | ((uint32_t)(py & 0xFF) << 16) | (valid ? 0x80000000u : 0u);
        const uint32_t *d = (const uint32_t *)(cdesc + ((size_t)f * cap + k) * 32);
#pragma unroll
        for (int w = 0; w < 8; w++) kdesc[k * 8 + w] = d[w];
        lock_a[k] = 0x7FFFFFFF; owner[k] = -1;
    }
    for (int i = tid; i < SGX_HISTO; i += NT) { hist[i] = 0; bad[i] = 0; }
    for (int i = tid; i < SGX_GRID_COLS * SGX_GRID_ROWS; i += NT) { cell_start[i] = 0; cell_fill[i] = 0; }
    if (tid == 0) { s_total = 0; s_rejected = 0; }
    SGX_THREADS_END
    SGX_SYNC();
    // ---- the 64x48 grid index (only an accelerator here: candidate order is carried by the preference key)
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < Nc; k += NT) { const uint32_t inf = kinfo[k]; if (inf & 0x80000000u) sgx_atomic_add(&cell_start[((inf >> 8) & 0xFF) * SGX_GRID_ROWS + ((inf >> 16) & 0xFF)], 1); }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    sgx_block_exclusive_scan_i32(cell_start, SGX_GRID_COLS * SGX_GRID_ROWS, &s_ngrid, tid);
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) cell_start[SGX_GRID_COLS * SGX_GRID_ROWS] = s_ngrid;
    for (int k = tid; k < Nc; k += NT) {
        const uint32_t inf = kinfo[k];
        if (inf & 0x80000000u) { const int c = ((inf >> 8) & 0xFF) * SGX_GRID_ROWS + ((inf >> 16) & 0xFF); cell_list[cell_start[c] + sgx_atomic_add(&cell_fill[c], 1)] = (uint16_t)k; }
    }
    SGX_THREADS_END
    SGX_SYNC();

    // ---- frame-level quantities (:1342-1353), uniform
    float Rcw[3][3], tcw[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r][c] = Tc[4 * r + c]; tcw[r] = Tc[4 * r + 3]; }
    float twc[3];
    for (int i = 0; i < 3; i++) {              // -Rcw.t()*tcw: transpose flag -> generic gemm, double accumulation, alpha = -1
        double s = 0; for (int k = 0; k < 3; k++) s += (double)Rcw[k][i] * (double)tcw[k];
        twc[i] = (float)(s * -1.0);
    }
    const float Rl2[3] = { Tl[8], Tl[9], Tl[10] };
    const float tlc2 = sgx_gemm3(Rl2, twc, Tl[11]);
    const float mb = cam.bf / cam.fx;
    const bool bForward = tlc2 > mb && !bMono, bBackward = -tlc2 > mb && !bMono;

    int *lock_cur = lock_a, *lock_new = lock_b;
    for (int sweep = 0; sweep < SGX_MATCH_CAP + 2; sweep++) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) s_changed = 0;
        for (int k = tid; k < Nc; k += NT) lock_new[k] = 0x7FFFFFFF;
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < Nl; i += NT) {
            int best = -1;
            const size_t li = (size_t)f * cap + i;
            if (l_has_mp[li] && !l_outlier[li]) {
                const float *xw = l_xw + 3 * li;
                const float x3x = sgx_gemm3(Rcw[0], xw, tcw[0]), x3y = sgx_gemm3(Rcw[1], xw, tcw[1]), x3z = sgx_gemm3(Rcw[2], xw, tcw[2]);
                const float invzc = (float)(1.0 / (double)x3z);
                const float u = cam.fx * x3x * invzc + cam.cx, v = cam.fy * x3y * invzc + cam.cy;
                if (!(invzc < 0) && !(u < cam.minX || u > cam.maxX) && !(v < cam.minY || v > cam.maxY)) {
                    const int lo = ((const int *)(lkeys_raw + li * 28))[5];
                    const float radius = th * sc.s[lo];
                    int minLevel, maxLevel;
                    if (bForward) { minLevel = lo; maxLevel = -1; } else if (bBackward) { minLevel = 0; maxLevel = lo; } else { minLevel = lo - 1; maxLevel = lo + 1; }
                    // GetFeaturesInArea cell window (Frame.cc:359-373)
                    const int c0x = max(0, (int)floorf((u - cam.minX - radius) * invW)), c1x = min(SGX_GRID_COLS - 1, (int)ceilf((u - cam.minX + radius) * invW));
                    const int c0y = max(0, (int)floorf((v - cam.minY - radius) * invH)), c1y = min(SGX_GRID_ROWS - 1, (int)ceilf((v - cam.minY + radius) * invH));
                    if (!(c0x >= SGX_GRID_COLS || c1x < 0 || c0y >= SGX_GRID_ROWS || c1y < 0)) {
                        const bool chk = (minLevel > 0) || (maxLevel >= 0);
                        const float ur = u - cam.bf * invzc;
                        const uint32_t *dmp = (const uint32_t *)(l_mpdesc + li * 32);
                        uint32_t dm[8];
#pragma unroll
                        for (int w = 0; w < 8; w++) dm[w] = dmp[w];
                        unsigned long long bestKey = ~0ull;
                        // cells (px, c0y..c1y) of one grid column are contiguous in the CSR index: one key range per column
                        for (int px = c0x; px <= c1x; px++)
                        for (int q = cell_start[px * SGX_GRID_ROWS + c0y], qe = cell_start[px * SGX_GRID_ROWS + c1y + 1]; q < qe; q++) {
                            const int k = cell_list[q];
                            const uint32_t inf = kinfo[k];
                            const int oct = inf & 0xFF, py = (inf >> 16) & 0xFF;
                            if (chk) { if (oct < minLevel) continue; if (maxLevel >= 0 && oct > maxLevel) continue; }
                            if (!(fabsf(kx[k] - u) < radius && fabsf(ky[k] - v) < radius)) continue;
                            if (lock_cur[k] < i) continue;                                   // holds an observed map point (:1407-1409)
                            if (kur[k] > 0) { if (fabsf(ur - kur[k]) > radius) continue; }    // :1411-1417
                            const int dist = sgx_hamming256(dm, &kdesc[k * 8]);
                            const unsigned long long key = ((unsigned long long)dist << 28) | ((unsigned long long)px << 22) | ((unsigned long long)py << 16) | (unsigned long long)k;
                            if (key < bestKey) bestKey = key;
                        }
                        if (bestKey != ~0ull && (int)(bestKey >> 28) <= SGX_TH_HIGH) best = (int)(bestKey & 0xFFFF);   // :1430
                    }
                }
            }
            choice[i] = best;
            if (best >= 0 && l_obs[li] > 0) sgx_atomic_min_i32(&lock_new[best], i);
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int k = tid; k < Nc; k += NT) if (lock_new[k] != lock_cur[k]) s_changed = 1;
        SGX_THREADS_END
        SGX_SYNC();
        int *t = lock_cur; lock_cur = lock_new; lock_new = t;
        if (!s_changed) break;
    }

    // ---- assignments, rotation histogram
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < Nl; i += NT) {
        const int k = choice[i];
        if (k < 0) continue;
        sgx_atomic_max(&owner[k], i);
        sgx_atomic_add(&s_total, 1);
        if (check_ori) {
            const float la = ((const float *)(lkeys_raw + ((size_t)f * cap + i) * 28))[3];
            float rot = la - kang[k];
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)round((double)(rot * (SGX_HISTO / 360.0f)));
            if (bin == SGX_HISTO) bin = 0;
            sgx_atomic_add(&hist[bin], 1);
            lock_new[i] = bin;                  // lock_new is free now: reuse as per-map-point bin
        }
    }
    SGX_THREADS_END
    SGX_SYNC();
    if (check_ori) {
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
        for (int i = tid; i < Nl; i += NT) {
            const int k = choice[i];
            if (k >= 0 && bad[lock_new[i]]) { sgx_atomic_add(&s_rejected, 1); lock_cur[k] = -2; }   // lock_cur reused as kill flag (-2)
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < Nc; k += NT) cur_match[(size_t)f * cap + k] = (check_ori && lock_cur[k] == -2) ? -1 : owner[k];
    for (int k = Nc + tid; k < cap; k += NT) cur_match[(size_t)f * cap + k] = -1;
    if (tid == 0) nmatches_out[f] = s_total - s_rejected;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_stereo_from_rgbd: Frame::ComputeStereoFromRGBD (Frame.cc:893-914) fused with the depth
// conversion imDepth.convertTo(CV_32F, 1/DepthMapFactor) (Tracking.cc:229-230) evaluated only at
// the keypoints.  grid = (ceil(cap/256), batch)
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_stereo_from_rgbd(int cap, const uint8_t *keys_raw, const int *n, const uint16_t *depth, int W, int H,
                                   float depth_factor_inv, float bf, float *uright, float *zdepth)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, i = (int)blockIdx.x * 256 + tid;
    if (i < cap) {
        float ur = -1.f, z = -1.f;
        if (i < n[f]) {
            const float *kp = (const float *)(keys_raw + ((size_t)f * cap + i) * 28);
            const int u = (int)kp[0], v = (int)kp[1];                    // cv::Mat::at<float>(float,float) truncates
            const float d = (float)depth[((size_t)f * H + v) * W + u] * depth_factor_inv;
            if (d > 0) { z = d; ur = kp[0] - bf / d; }
        }
        uright[(size_t)f * cap + i] = ur; zdepth[(size_t)f * cap + i] = z;
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_unproject: Frame::UnprojectStereo (Frame.cc:916-930) for every keypoint with depth > 0:
// x3Dw = mRwc*x3Dc + mOw, mRwc = Rcw^T, mOw = -Rcw^T tcw (Frame.cc:288-294).  has[i] = depth>0.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_unproject(int cap, const uint8_t *keys_raw, const int *n, const float *zdepth, const float *Tcw, SgxCam cam,
                            float *xw, uint8_t *has)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, i = (int)blockIdx.x * 256 + tid;
    if (i < cap) {
        const size_t o = (size_t)f * cap + i;
        uint8_t h = 0;
        float X[3] = {0.f, 0.f, 0.f};
        if (i < n[f]) {
            const float z = zdepth[o];
            if (z > 0) {
                const float *T = Tcw + 16 * f;
                const float *kp = (const float *)(keys_raw + o * 28);
                const float invfx = 1.0f / cam.fx, invfy = 1.0f / cam.fy;
                const float xc[3] = { (kp[0] - cam.cx) * z * invfx, (kp[1] - cam.cy) * z * invfy, z };
                for (int r = 0; r < 3; r++) {
                    double s = 0; for (int k = 0; k < 3; k++) s += (double)T[4 * k + r] * (double)T[4 * k + 3];
                    const float Ow = (float)(s * -1.0);
                    const float Rrow[3] = { T[r], T[4 + r], T[8 + r] };
                    X[r] = sgx_gemm3(Rrow, xc, Ow);
                }
                h = 1;
            }
        }
        xw[3 * o] = X[0]; xw[3 * o + 1] = X[1]; xw[3 * o + 2] = X[2]; has[o] = h;
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_motion_model: the caller-side constant-velocity prediction of Tracking (Tracking.cc:463-470, :914):
//   LastTwc = [Rlw^T | -Rlw^T tlw],  mVelocity = Tcw_cur * LastTwc,  predicted = mVelocity * Tcw_cur
// i.e. pred = Tcur * inv(Tprev) * Tcur for the next frame.  4x4 float products as cv::gemm's 4x4 path
// (float dot, left to right).  One thread per frame.
// ---------------------------------------------------------------------------------------------
SGX_DEV void sgx_mat4_mul(const float *A, const float *B, float *C)
{
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        const float t = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j] + A[4 * i + 3] * B[12 + j];
        C[4 * i + j] = (float)((double)t * 1.0);
    }
}

SGX_KERNEL(64) k_motion_model(int batch, const float *Tcur, const float *Tprev, const uint8_t *valid, float *Tpred)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.x * 64 + tid;
    if (f < batch) {
        const float *Tc = Tcur + 16 * f, *Tp = Tprev + 16 * f;
        float out[16];
        if (valid && !valid[f]) { for (int i = 0; i < 16; i++) out[i] = Tc[i]; }
        else {
            float Twc[16];
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) Twc[4 * r + c] = Tp[4 * c + r];
                double s = 0; for (int k = 0; k < 3; k++) s += (double)Tp[4 * k + r] * (double)Tp[4 * k + 3];   // mOw = -Rcw^T tcw (Frame.cc:288-294)
                Twc[4 * r + 3] = (float)(s * -1.0);
            }
            Twc[12] = 0.f; Twc[13] = 0.f; Twc[14] = 0.f; Twc[15] = 1.f;
            float V[16]; sgx_mat4_mul(Tc, Twc, V);
            sgx_mat4_mul(V, Tc, out);
        }
        for (int i = 0; i < 16; i++) Tpred[16 * f + i] = out[i];
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_match_project_local: Tracking::SearchLocalPoints' inner work (Tracking.cc:1284-1311) for one frame:
//   Frame::isInFrustum(pMP, 0.5)            Frame.cc:296-352   (incl. MapPoint::PredictScale, MapPoint.cc:402-417)
//   ORBmatcher(0.8).SearchByProjection(F, vpMapPoints, th)   ORBmatcher.cc:45-129 (RadiusByViewingCos :131-137)
// One 1024-thread workgroup per frame, thread i <-> local map point i (strided).  Same ingredients as
// k_match_project_frame: LDS-staged frame + CSR grid, preference keys, lock table + Jacobi sweeps for the greedy
// "a keypoint that holds an observed map point is skipped" rule (:87-89).  Here the best AND the second-best available
// candidates matter (ratio test :120-121): second = 2nd smallest (distance, scan order) key among the available candidates.
// cur_mp_obs[k] > 0 marks keypoints already holding an observed map point (motion-model stage): locked from the start.
// Outputs: cur_match[k] = local map point newly assigned to keypoint k (last writer in index order) or -1; in_view = mbTrackInView.
// ---------------------------------------------------------------------------------------------
#define SGX_LOCAL_CAP 4096
#define SGX_LOCAL_SLOTS (SGX_LOCAL_CAP / SGX_MATCH_THREADS)      /* local map points per thread */
#define SGX_LOCAL_KEEP 4                                          /* smallest candidate keys kept per map point */

struct SgxLocalProj { float u, v, radius, projXR; int lvl; bool ok; };

// Frame::isInFrustum + MapPoint::PredictScale + the search radius of SearchByProjection for one local map point
SGX_DEV SgxLocalProj sgx_local_project(const float *P, const float *Pn, float max_dist, float min_dist, const float (*Rcw)[3], const float *tcw, const float *Ow,
                                       const SgxCam &cam, const SgxScales &sc, int nlevels, float log_scale_factor, float th, float viewing_cos_limit)
{
    SgxLocalProj r; r.lvl = 0; r.radius = 0.f; r.projXR = 0.f;
    const float pcx = sgx_gemm3(Rcw[0], P, tcw[0]), pcy = sgx_gemm3(Rcw[1], P, tcw[1]), pcz = sgx_gemm3(Rcw[2], P, tcw[2]);
    bool ok = !(pcz < 0.0f);
    const float invz = 1.0f / pcz;
    r.u = cam.fx * pcx * invz + cam.cx; r.v = cam.fy * pcy * invz + cam.cy;
    ok = ok && !(r.u < cam.minX || r.u > cam.maxX) && !(r.v < cam.minY || r.v > cam.maxY);
    const float maxDistance = 1.2f * max_dist, minDistance = 0.8f * min_dist;                     // MapPoint.cc:372-383
    const float po0 = P[0] - Ow[0], po1 = P[1] - Ow[1], po2 = P[2] - Ow[2];
    const float dist = (float)sqrt((double)po0 * po0 + (double)po1 * po1 + (double)po2 * po2);   // cv::norm accumulates in double
    ok = ok && !(dist < minDistance || dist > maxDistance);
    const float viewCos = (float)(((double)po0 * Pn[0] + (double)po1 * Pn[1] + (double)po2 * Pn[2]) / (double)dist);
    ok = ok && !(viewCos < viewing_cos_limit);
    r.ok = ok;
    if (ok) {
        int lvl = (int)ceilf((float)log((double)(max_dist / dist)) / log_scale_factor);           // MapPoint::PredictScale (logf in the reference)
        if (lvl < 0) lvl = 0; else if (lvl >= nlevels) lvl = nlevels - 1;
        r.lvl = lvl;
        r.projXR = r.u - cam.bf * invz;
        float rad = viewCos > 0.998 ? 2.5f : 4.0f;                                                // RadiusByViewingCos
        if (th != 1.0f) rad *= th;
        r.radius = rad * sc.s[lvl];
    }
    return r;
}

// candidate key: (Hamming distance, scan order = grid column, grid row, keypoint index) — smaller = preferred / earlier in the reference's scan
#define SGX_LKEY(dd, px, py, k) (((uint32_t)(dd) << 23) | ((uint32_t)(px) << 17) | ((uint32_t)(py) << 11) | (uint32_t)(k))
#define SGX_LKEY_NONE 0xFFFFFFFFu

SGX_KERNEL(SGX_MATCH_THREADS) k_match_project_local(
    int cap, const uint8_t *ckeys_raw, const uint8_t *cdesc, const float *curight, const int *cn, const float *cTcw, const int *cur_mp_obs,
    int mcap, const int *mn, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc,
    const int *m_obs, const uint8_t *m_skip,
    SgxCam cam, SgxScales sc, int nlevels, float log_scale_factor, float th, float nnratio, float viewing_cos_limit,
    int *cur_match, int *nmatches_out, uint8_t *in_view)
{
    SGX_LDS float kx[SGX_MATCH_CAP], ky[SGX_MATCH_CAP], kur[SGX_MATCH_CAP];
    SGX_LDS uint32_t kinfo[SGX_MATCH_CAP];
    SGX_LDS uint32_t kdesc[SGX_MATCH_CAP * 8];
    SGX_LDS int lock_a[SGX_MATCH_CAP], lock_b[SGX_MATCH_CAP];
    SGX_LDS int owner[SGX_MATCH_CAP];
    SGX_LDS int cell_start[SGX_GRID_COLS * SGX_GRID_ROWS + 1];
    SGX_LDS int cell_fill[SGX_GRID_COLS * SGX_GRID_ROWS];
    SGX_LDS uint16_t cell_list[SGX_MATCH_CAP];
    SGX_LDS uint16_t choice[SGX_LOCAL_CAP];           // 0xFFFF = none
    SGX_LDS int s_changed, s_total, s_ngrid;
    // per-thread state that lives across the sweeps: the SGX_LOCAL_KEEP smallest candidate keys of each of the thread's map points
    // (lock-independent: the locks only ever REMOVE candidates), + bit 0 = list truncated, bit 1 = the point's Observations() > 0
    SGX_PRIV_DECL(uint32_t, ckey, SGX_LOCAL_SLOTS * SGX_LOCAL_KEEP, SGX_MATCH_THREADS);
    SGX_PRIV_DECL(uint32_t, cflag, SGX_LOCAL_SLOTS, SGX_MATCH_THREADS);

    const int f = (int)blockIdx.x;
    SGX_WAVE_PRIORITY(3);          // latency-critical one-workgroup-per-frame kernel (see k_pose_opt)
    const int Nc = min(cn[f], cap), Nm = min(min(mn[f], mcap), SGX_LOCAL_CAP);
    const int NT = (int)blockDim.x;
    const float *Tc = cTcw + 16 * f;
    const float invW = (float)SGX_GRID_COLS / (cam.maxX - cam.minX), invH = (float)SGX_GRID_ROWS / (cam.maxY - cam.minY);

    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < Nc; k += NT) {
        const float *kp = (const float *)(ckeys_raw + ((size_t)f * cap + k) * 28);
        const float x = kp[0], y = kp[1];
        kx[k] = x; ky[k] = y; kur[k] = curight[(size_t)f * cap + k];
        const int oct = ((const int *)kp)[5];
        const int px = (int)round((double)((x - cam.minX) * invW)), py = (int)round((double)((y - cam.minY) * invH));
        const bool valid = !(px < 0 || px >= SGX_GRID_COLS || py < 0 || py >= SGX_GRID_ROWS);
        kinfo[k] = (uint32_t)(oct & 0xFF) | ((uint32_t)(px & 0xFF) << 8) | ((uint32_t)(py & 0xFF) << 16) | (valid ? 0x80000000u : 0u);
        const uint32_t *d = (const uint32_t *)(cdesc + ((size_t)f * cap + k) * 32);
#pragma unroll
        for (int w = 0; w < 8; w++) kdesc[k * 8 + w] = d[w];
        // a keypoint holding an observed map point is unavailable to every local map point: lock index -1
        lock_a[k] = (cur_mp_obs && cur_mp_obs[(size_t)f * cap + k] > 0) ? -1 : 0x7FFFFFFF;
        owner[k] = -1;
    }
    for (int i = tid; i < SGX_GRID_COLS * SGX_GRID_ROWS; i += NT) { cell_start[i] = 0; cell_fill[i] = 0; }
    if (tid == 0) s_total = 0;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < Nc; k += NT) { const uint32_t inf = kinfo[k]; if (inf & 0x80000000u) sgx_atomic_add(&cell_start[((inf >> 8) & 0xFF) * SGX_GRID_ROWS + ((inf >> 16) & 0xFF)], 1); }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    sgx_block_exclusive_scan_i32(cell_start, SGX_GRID_COLS * SGX_GRID_ROWS, &s_ngrid, tid);
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) cell_start[SGX_GRID_COLS * SGX_GRID_ROWS] = s_ngrid;
    for (int k = tid; k < Nc; k += NT) {
        const uint32_t inf = kinfo[k];
        if (inf & 0x80000000u) { const int c = ((inf >> 8) & 0xFF) * SGX_GRID_ROWS + ((inf >> 16) & 0xFF); cell_list[cell_start[c] + sgx_atomic_add(&cell_fill[c], 1)] = (uint16_t)k; }
    }
    SGX_THREADS_END
    SGX_SYNC();

    float Rcw[3][3], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r][c] = Tc[4 * r + c]; tcw[r] = Tc[4 * r + 3]; }
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)Rcw[k][i] * (double)tcw[k]; Ow[i] = (float)(s * -1.0); }   // mOw, Frame.cc:288-294

    // ---- build: project every local map point once, keep its smallest candidate keys (no lock filter)
    SGX_THREADS_BEGIN(tid)
    SGX_PRIV_BIND(ckey, tid); SGX_PRIV_BIND(cflag, tid);
#pragma unroll
    for (int j = 0; j < SGX_LOCAL_SLOTS; j++) {
        const int i = tid + j * SGX_MATCH_THREADS;
        uint32_t c0 = SGX_LKEY_NONE, c1 = SGX_LKEY_NONE, c2 = SGX_LKEY_NONE, c3 = SGX_LKEY_NONE, flag = 0;
        if (i < Nm) {
            const size_t mi = (size_t)f * mcap + i;
            uint8_t vis = 0;
            if (!m_skip[mi]) {
                const SgxLocalProj pr = sgx_local_project(m_xw + 3 * mi, m_normal + 3 * mi, m_max_dist[mi], m_min_dist[mi], Rcw, tcw, Ow, cam, sc, nlevels,
                                                          log_scale_factor, th, viewing_cos_limit);
                if (pr.ok) {
                    vis = 1;
                    if (m_obs[mi] > 0) flag |= 2u;
                    const float u = pr.u, v = pr.v, radius = pr.radius;
                    const int minLevel = pr.lvl - 1, maxLevel = pr.lvl;
                    const int c0x = max(0, (int)floorf((u - cam.minX - radius) * invW)), c1x = min(SGX_GRID_COLS - 1, (int)ceilf((u - cam.minX + radius) * invW));
                    const int c0y = max(0, (int)floorf((v - cam.minY - radius) * invH)), c1y = min(SGX_GRID_ROWS - 1, (int)ceilf((v - cam.minY + radius) * invH));
                    if (!(c0x >= SGX_GRID_COLS || c1x < 0 || c0y >= SGX_GRID_ROWS || c1y < 0)) {
                        const uint32_t *dmp = (const uint32_t *)(m_desc + mi * 32);
                        uint32_t dm[8];
#pragma unroll
                        for (int w = 0; w < 8; w++) dm[w] = dmp[w];
                        int cnt = 0;
                        for (int px = c0x; px <= c1x; px++)
                        for (int q = cell_start[px * SGX_GRID_ROWS + c0y], qe = cell_start[px * SGX_GRID_ROWS + c1y + 1]; q < qe; q++) {
                            const int k = cell_list[q];
                            const uint32_t inf = kinfo[k];
                            const int oct = inf & 0xFF, py = (inf >> 16) & 0xFF;
                            if (oct < minLevel || oct > maxLevel) continue;        // bCheckLevels holds (maxLevel >= 0)
                            if (!(fabsf(kx[k] - u) < radius && fabsf(ky[k] - v) < radius)) continue;
                            if (kur[k] > 0) { if (fabsf(pr.projXR - kur[k]) > radius) continue; }
                            const int dd = sgx_hamming256(dm, &kdesc[k * 8]);
                            uint32_t key = SGX_LKEY(dd, px, py, k);
                            cnt++;
                            if (key < c3) {                                            // sorted insertion, static registers
                                c3 = key;
                                if (c3 < c2) { const uint32_t t = c2; c2 = c3; c3 = t; }
                                if (c2 < c1) { const uint32_t t = c1; c1 = c2; c2 = t; }
                                if (c1 < c0) { const uint32_t t = c0; c0 = c1; c1 = t; }
                            }
                        }
                        if (cnt > SGX_LOCAL_KEEP) flag |= 1u;
                    }
                }
            }
            in_view[mi] = vis;
        }
        ckey[j * SGX_LOCAL_KEEP + 0] = c0; ckey[j * SGX_LOCAL_KEEP + 1] = c1; ckey[j * SGX_LOCAL_KEEP + 2] = c2; ckey[j * SGX_LOCAL_KEEP + 3] = c3;
        cflag[j] = flag;
    }
    SGX_THREADS_END
    SGX_SYNC();

    // ---- Jacobi sweeps over the lock table: each map point takes its two smallest keys whose keypoint is not locked by a smaller index
    int *lock_cur = lock_a, *lock_new = lock_b;
    for (int sweep = 0; sweep < SGX_LOCAL_CAP + 2; sweep++) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) s_changed = 0;
        for (int k = tid; k < Nc; k += NT) lock_new[k] = (cur_mp_obs && cur_mp_obs[(size_t)f * cap + k] > 0) ? -1 : 0x7FFFFFFF;
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        SGX_PRIV_BIND(ckey, tid); SGX_PRIV_BIND(cflag, tid);
#pragma unroll
        for (int j = 0; j < SGX_LOCAL_SLOTS; j++) {
            const int i = tid + j * SGX_MATCH_THREADS;
            if (i < Nm) {
                uint32_t k1 = SGX_LKEY_NONE, k2 = SGX_LKEY_NONE;
#pragma unroll
                for (int q = 0; q < SGX_LOCAL_KEEP; q++) {
                    const uint32_t key = ckey[j * SGX_LOCAL_KEEP + q];
                    if (key != SGX_LKEY_NONE && !(lock_cur[key & 0x7FF] < i)) { if (k1 == SGX_LKEY_NONE) k1 = key; else if (k2 == SGX_LKEY_NONE) k2 = key; }
                }
                if ((cflag[j] & 1u) && k2 == SGX_LKEY_NONE) {
                    // rare: the kept list was truncated and fewer than two of its keypoints are still available -> full rescan with the lock filter
                    const size_t mi = (size_t)f * mcap + i;
                    const SgxLocalProj pr = sgx_local_project(m_xw + 3 * mi, m_normal + 3 * mi, m_max_dist[mi], m_min_dist[mi], Rcw, tcw, Ow, cam, sc, nlevels,
                                                              log_scale_factor, th, viewing_cos_limit);
                    const float u = pr.u, v = pr.v, radius = pr.radius;
                    const int minLevel = pr.lvl - 1, maxLevel = pr.lvl;
                    const int c0x = max(0, (int)floorf((u - cam.minX - radius) * invW)), c1x = min(SGX_GRID_COLS - 1, (int)ceilf((u - cam.minX + radius) * invW));
                    const int c0y = max(0, (int)floorf((v - cam.minY - radius) * invH)), c1y = min(SGX_GRID_ROWS - 1, (int)ceilf((v - cam.minY + radius) * invH));
                    const uint32_t *dm = (const uint32_t *)(m_desc + mi * 32);
                    k1 = SGX_LKEY_NONE; k2 = SGX_LKEY_NONE;
                    for (int px = c0x; px <= c1x; px++)
                    for (int q = cell_start[px * SGX_GRID_ROWS + c0y], qe = cell_start[px * SGX_GRID_ROWS + c1y + 1]; q < qe; q++) {
                        const int k = cell_list[q];
                        const uint32_t inf = kinfo[k];
                        const int oct = inf & 0xFF, py = (inf >> 16) & 0xFF;
                        if (oct < minLevel || oct > maxLevel) continue;
                        if (!(fabsf(kx[k] - u) < radius && fabsf(ky[k] - v) < radius)) continue;
                        if (lock_cur[k] < i) continue;
                        if (kur[k] > 0) { if (fabsf(pr.projXR - kur[k]) > radius) continue; }
                        const uint32_t key = SGX_LKEY(sgx_hamming256(dm, &kdesc[k * 8]), px, py, k);
                        if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
                    }
                }
                int best = -1;
                if (k1 != SGX_LKEY_NONE) {
                    const int bestDist = (int)(k1 >> 23), bestLevel = (int)(kinfo[k1 & 0x7FF] & 0xFF);
                    const int bestDist2 = k2 != SGX_LKEY_NONE ? (int)(k2 >> 23) : 256, bestLevel2 = k2 != SGX_LKEY_NONE ? (int)(kinfo[k2 & 0x7FF] & 0xFF) : -1;
                    if (bestDist <= SGX_TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2)) best = (int)(k1 & 0x7FF);
                }
                choice[i] = best >= 0 ? (uint16_t)best : (uint16_t)0xFFFF;
                if (best >= 0 && (cflag[j] & 2u)) sgx_atomic_min_i32(&lock_new[best], i);
            }
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int k = tid; k < Nc; k += NT) if (lock_new[k] != lock_cur[k]) s_changed = 1;
        SGX_THREADS_END
        SGX_SYNC();
        int *t = lock_cur; lock_cur = lock_new; lock_new = t;
        if (!s_changed) break;
    }

    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < Nm; i += NT) { const int k = choice[i]; if (k != 0xFFFF) { sgx_atomic_max(&owner[k], i); sgx_atomic_add(&s_total, 1); } }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < cap; k += NT) cur_match[(size_t)f * cap + k] = k < Nc ? owner[k] : -1;
    if (tid == 0) nmatches_out[f] = s_total;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_make_map_points: MapPoint::MapPoint(Pos, pMap, pFrame, idxF) (MapPoint.cc:45-67) for every keypoint of a frame that has
// depth — the "visual odometry" points Tracking::UpdateLastFrame creates (Tracking.cc:840-904): normal = (P - Ow)/|P - Ow|,
// mfMaxDistance = dist * scale[octave], mfMinDistance = mfMaxDistance / scale[nlevels-1], descriptor = the keypoint's.
// Records go to slice `half` of a per-frame local-map ring of 2*cap entries; entries without depth are marked skip.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_make_map_points(int cap, int half, const uint8_t *keys_raw, const int *n, const float *xw, const uint8_t *has, const uint8_t *desc,
                                  const float *Tcw, SgxScales sc, int nlevels, float *m_xw, float *m_normal, float *m_min, float *m_max, uint8_t *m_desc, uint8_t *m_skip)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, i = (int)blockIdx.x * 256 + tid;
    if (i < cap) {
        const size_t o = (size_t)f * cap + i, m = (size_t)f * 2 * cap + (size_t)half * cap + i;
        const bool ok = i < n[f] && has[o];
        m_skip[m] = ok ? 0 : 1;
        if (ok) {
            const float *T = Tcw + 16 * f;
            float P[3], PO[3];
            for (int r = 0; r < 3; r++) {
                double s = 0; for (int k = 0; k < 3; k++) s += (double)T[4 * k + r] * (double)T[4 * k + 3];
                P[r] = xw[3 * o + r]; PO[r] = P[r] - (float)(s * -1.0);
            }
            const double nrm = sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
            const float dist = (float)nrm, inv = (float)(1.0 / nrm);
            const int oct = ((const int *)(keys_raw + o * 28))[5];
            const float mx = dist * sc.s[oct];
            for (int r = 0; r < 3; r++) { m_xw[3 * m + r] = P[r]; m_normal[3 * m + r] = PO[r] * inv; }
            m_max[m] = mx; m_min[m] = mx / sc.s[nlevels - 1];
            const uint32_t *d = (const uint32_t *)(desc + o * 32); uint32_t *dd = (uint32_t *)(m_desc + m * 32);
#pragma unroll
            for (int w = 0; w < 8; w++) dd[w] = d[w];
        }
    }
    SGX_THREADS_END
}

// k_merge_matches: the frame's mvpMapPoints after TrackWithMotionModel + SearchLocalPoints as ONE index into a combined
// map-point table [last frame's points (cap) | local-map ring (2*cap)]: a local-map match replaces a visual-odometry point
// (Observations()==0 points are overwritten, ORBmatcher.cc:87-89,123); motion-model outliers were dropped (Tracking.cc:941-956).
SGX_KERNEL(256) k_merge_matches(int cap, const int *n, const int *match_last, const uint8_t *outlier_last, const int *match_local, int *merged, int *cur_mp_obs)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, i = (int)blockIdx.x * 256 + tid;
    if (i < cap) {
        const size_t o = (size_t)f * cap + i;
        int idx = -1;
        if (i < n[f]) {
            if (match_local && match_local[o] >= 0) idx = cap + match_local[o];
            else if (match_last[o] >= 0 && !outlier_last[o]) idx = match_last[o];
        }
        if (merged) merged[o] = idx;
        if (cur_mp_obs) cur_mp_obs[o] = (i < n[f] && match_last[o] >= 0 && !outlier_last[o]) ? 0 : -1;    // VO points: Observations() == 0
    }
    SGX_THREADS_END
}

// k_gather_xw: combined map-point position table for pose optimisation #2: [last.xw (cap) | local-map xw (2*cap)] per frame
SGX_KERNEL(256) k_gather_xw(int cap, const float *xw_last, const float *m_xw, float *xw_all)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, i = (int)blockIdx.x * 256 + tid;
    if (i < 3 * cap) {
        const float *src = i < cap ? xw_last + 3 * ((size_t)f * cap + i) : m_xw + 3 * ((size_t)f * 2 * cap + (i - cap));
        float *dst = xw_all + 3 * ((size_t)f * 3 * cap + i);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_gray_from_color: the cvtColor at the top of Tracking::GrabImageRGBD (Tracking.cc:214-227): CV_RGB2GRAY / CV_BGR2GRAY (3 channels) or
// CV_RGBA2GRAY / CV_BGRA2GRAY (4 channels) on 8-bit images, OpenCV's fixed-point form
//   gray = (R*4899 + G*9617 + B*1868 + (1 << 13)) >> 14            (R2Y, G2Y, B2Y at yuv_shift 14, rounding CV_DESCALE)
// One thread per 4 output pixels: 3 (or 4) aligned dword loads, one dword store.  grid = (ceil(W/4/64), H, B), block 64.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(64) k_gray_from_color(int W, int H, const uint8_t *src, int src_pitch, int channels, int blue_first, uint8_t *dst, int dst_pitch)
{
    SGX_THREADS_BEGIN(tid)
    const int x4 = ((int)blockIdx.x * 64 + tid) * 4, y = (int)blockIdx.y, b = (int)blockIdx.z;
    if (x4 < W) {
        const uint8_t *row = src + ((size_t)b * H + y) * src_pitch + (size_t)x4 * channels;
        uint32_t out = 0;
        const int c0 = blue_first ? 1868 : 4899, c2 = blue_first ? 4899 : 1868;
        uint32_t w[4];
        const int nd = channels;                               // 4 pixels = `channels` dwords
        for (int i = 0; i < 4; i++) w[i] = (i < nd && x4 + 4 <= W) ? ((const uint32_t *)row)[i] : 0u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (x4 + i < W) {
                int p0, p1, p2;
                if (x4 + 4 <= W) {
                    const int o = i * channels;                // byte offset of the pixel inside the 4-pixel group
                    p0 = (w[o >> 2] >> (8 * (o & 3))) & 255; p1 = (w[(o + 1) >> 2] >> (8 * ((o + 1) & 3))) & 255; p2 = (w[(o + 2) >> 2] >> (8 * ((o + 2) & 3))) & 255;
                } else { p0 = row[i * channels]; p1 = row[i * channels + 1]; p2 = row[i * channels + 2]; }     // ragged row end: byte loads
                const int v = (p0 * c0 + p1 * 9617 + p2 * c2 + (1 << 13)) >> 14;
                out |= (uint32_t)v << (8 * i);
            }
        }
        uint8_t *d = dst + ((size_t)b * H + y) * dst_pitch + x4;
        if (x4 + 4 <= W) *(uint32_t *)d = out;
        else for (int i = 0; x4 + i < W; i++) d[i] = (uint8_t)(out >> (8 * i));
    }
    SGX_THREADS_END
}

// k_flow_affine: harness stand-in for the LK tracker on synthetic streams (bench.py / tests; see sg_slam_amd/synth.py flow_affine): prev = A * (x, y, 1) per
// keypoint, optionally displaced by `shift` inside the frame's first box (an independently moving object).  Not part of the reference path.
SGX_KERNEL(256) k_flow_affine(int cap, const uint8_t *keys_raw, const int *n, const float *A, const float *shift, const float *boxes, int max_boxes, float *prev_xy)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, i = (int)blockIdx.x * 256 + tid;
    if (i < cap) {
        const float *kp = (const float *)(keys_raw + ((size_t)f * cap + i) * 28);
        const float x = kp[0], y = kp[1];
        const float *a = A + 6 * f;
        float px = a[0] * x + a[1] * y + a[2], py = a[3] * x + a[4] * y + a[5];
        if (shift && boxes) {
            const float *bx = boxes + 4 * (size_t)f * max_boxes;
            if (x > bx[0] && x < bx[0] + bx[2] && y > bx[1] && y < bx[1] + bx[3]) { px += shift[2 * f]; py += shift[2 * f + 1]; }
        }
        prev_xy[2 * ((size_t)f * cap + i)] = px; prev_xy[2 * ((size_t)f * cap + i) + 1] = py;
    }
    SGX_THREADS_END
}
