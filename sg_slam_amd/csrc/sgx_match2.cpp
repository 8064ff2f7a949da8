// sgx_match2.cpp — host side of the LocalMapping matcher gates (include/sgx.h): sgx_hamming_matrix, sgx_match_search_for_triangulation.
// Reference behaviour: src/sg-slam/src/ORBmatcher.cc:659-827, :1649-1665.
#include "sgx_match2_kernels.h"
#include "sgx_stage.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

extern "C" int sgx_hamming_matrix_dev(const uint8_t *d_desc_a, int na, const uint8_t *d_desc_b, int nb, uint16_t *d_out, void *stream)
{
    if (na < 0 || nb < 0 || (na > 0 && !d_desc_a) || (nb > 0 && !d_desc_b) || (na > 0 && nb > 0 && !d_out)) return SGX_ERR_INVALID;
    if (na == 0 || nb == 0) return SGX_OK;
    SGX_LAUNCH(k_hamming_matrix, dim3((nb + 15) / 16, (na + 15) / 16), dim3(256), (sgx_stream_t)stream, (const uint32_t *)d_desc_a, na, (const uint32_t *)d_desc_b, nb, d_out);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_hamming_matrix(const uint8_t *desc_a, int na, const uint8_t *desc_b, int nb, uint16_t *out)
{
    if (na < 0 || nb < 0 || (na > 0 && !desc_a) || (nb > 0 && !desc_b) || (na > 0 && nb > 0 && !out)) return SGX_ERR_INVALID;
    if (na == 0 || nb == 0) return SGX_OK;
    SgxStaged dA, dB, dO; int rc;
    if ((rc = dA.put(0, desc_a, (size_t)na * 32)) != SGX_OK || (rc = dB.put(1, desc_b, (size_t)nb * 32)) != SGX_OK || (rc = dO.put(2, nullptr, (size_t)na * nb * 2)) != SGX_OK) return rc;
    if ((rc = sgx_hamming_matrix_dev((const uint8_t *)dA.p, na, (const uint8_t *)dB.p, nb, (uint16_t *)dO.p, nullptr)) != SGX_OK) return rc;
    SGX_CHECK_HIP(hipMemcpy(out, dO.p, (size_t)na * nb * 2, hipMemcpyDeviceToHost));
    return SGX_OK;
}

// FeatureVector (std::map<NodeId, std::vector<unsigned>>) of one keyframe from the per-feature node ids: nodes ascending, feature indices ascending inside a node
static void group_by_node(const int32_t *node, int n, std::vector<int> &items, std::vector<int> &ids, std::vector<int> &start)
{
    items.clear(); ids.clear(); start.clear();
    for (int i = 0; i < n; i++) if (node[i] >= 0) items.push_back(i);
    std::stable_sort(items.begin(), items.end(), [&](int x, int y) { return node[x] < node[y]; });
    for (size_t q = 0; q < items.size(); q++) if (q == 0 || node[items[q]] != node[items[q - 1]]) { ids.push_back(node[items[q]]); start.push_back((int)q); }
    start.push_back((int)items.size());
}

// keypoint octaves index the per-level tables of the kernel arguments (12 entries): reject frames whose octaves fall outside the extractor's levels
static bool octaves_ok(const sgx_keypoint *k, int n, int nlevels)
{
    for (int i = 0; i < n; i++) if (k[i].octave < 0 || k[i].octave >= nlevels) return false;
    return true;
}

extern "C" int sgx_match_search_for_triangulation(
    int n1, const sgx_keypoint *keys1_un, const uint8_t *desc1, const float *uright1, const uint8_t *has_mp1, const int32_t *feat_node1, const float *cam_center1,
    int n2, const sgx_keypoint *keys2_un, const uint8_t *desc2, const float *uright2, const uint8_t *has_mp2, const int32_t *feat_node2, const float *Tcw2,
    const float *F12, const sgx_camera *cam2, const float *scale_factors2, const float *level_sigma2_2, int nlevels, int only_stereo, int check_orientation,
    int32_t *pairs, int32_t *npairs)
{
    if (n1 < 0 || n2 < 0 || !npairs || !F12 || !cam2 || !scale_factors2 || !level_sigma2_2 || nlevels < 1 || nlevels > 12 || !cam_center1 || !Tcw2) return SGX_ERR_INVALID;
    *npairs = 0;
    if (n1 == 0 || n2 == 0) return SGX_OK;
    if (!keys1_un || !desc1 || !uright1 || !has_mp1 || !feat_node1 || !keys2_un || !desc2 || !uright2 || !has_mp2 || !feat_node2 || !pairs) return SGX_ERR_INVALID;
    if (!octaves_ok(keys1_un, n1, nlevels) || !octaves_ok(keys2_un, n2, nlevels)) return SGX_ERR_INVALID;
    std::vector<int> it1, id1, st1, it2, id2, st2, job;
    group_by_node(feat_node1, n1, it1, id1, st1); group_by_node(feat_node2, n2, it2, id2, st2);
    for (size_t a = 0, b = 0; a < id1.size() && b < id2.size();) {               // the lock-step walk of the two maps (:692-776)
        if (id1[a] == id2[b]) { job.push_back(st1[a]); job.push_back(st1[a + 1]); job.push_back(st2[b]); job.push_back(st2[b + 1]); a++; b++; }
        else if (id1[a] < id2[b]) a++; else b++;
    }
    SgxTriArgs A; memset(&A, 0, sizeof A);
    A.n1 = n1; A.n2 = n2; A.nnodes = (int)(job.size() / 4);
    // epipole of KF1's centre in KF2 (:666-674): cv::Mat float products, left to right
    {
        float C2[3];
        for (int r = 0; r < 3; r++) {           // cv::Mat C2 = R2w*Cw + t2w: one cv::gemm(alpha = 1, beta = 1): float dot left to right, then (float)(dot + t) in double
            const float *row = Tcw2 + 4 * r; const float t = row[0] * cam_center1[0] + row[1] * cam_center1[1] + row[2] * cam_center1[2];
            C2[r] = (float)((double)t * 1.0 + 1.0 * (double)row[3]);
        }
        const float invz = 1.0f / C2[2];
        A.ex = cam2->fx * C2[0] * invz + cam2->cx; A.ey = cam2->fy * C2[1] * invz + cam2->cy;
    }
    for (int i = 0; i < 9; i++) A.F12[i] = F12[i];
    for (int i = 0; i < nlevels; i++) { A.scale2.s[i] = scale_factors2[i]; A.sigma2_2.s[i] = level_sigma2_2[i]; }
    A.only_stereo = only_stereo; A.check_ori = check_orientation;
    SgxStaged b[16]; int rc;
    const int dummy = 0;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, keys1_un, (size_t)n1 * 28); PUT(1, desc1, (size_t)n1 * 32); PUT(2, uright1, (size_t)n1 * 4); PUT(3, has_mp1, (size_t)n1);
    PUT(4, keys2_un, (size_t)n2 * 28); PUT(5, desc2, (size_t)n2 * 32); PUT(6, uright2, (size_t)n2 * 4); PUT(7, has_mp2, (size_t)n2);
    PUT(8, it1.empty() ? &dummy : it1.data(), it1.empty() ? 4 : it1.size() * 4); PUT(9, it2.empty() ? &dummy : it2.data(), it2.empty() ? 4 : it2.size() * 4);
    PUT(10, job.empty() ? &dummy : job.data(), job.empty() ? 4 : job.size() * 4);
    PUT(11, nullptr, (size_t)n1 * 4); PUT(12, nullptr, (size_t)n2); PUT(13, nullptr, 4);
#undef PUT
    A.keys1 = (const uint8_t *)b[0].p; A.desc1 = (const uint32_t *)b[1].p; A.uright1 = (const float *)b[2].p; A.has_mp1 = (const uint8_t *)b[3].p;
    A.keys2 = (const uint8_t *)b[4].p; A.desc2 = (const uint32_t *)b[5].p; A.uright2 = (const float *)b[6].p; A.has_mp2 = (const uint8_t *)b[7].p;
    A.items1 = (const int *)b[8].p; A.items2 = (const int *)b[9].p; A.job = (const int *)b[10].p;
    A.match12 = (int *)b[11].p; A.matched2 = (uint8_t *)b[12].p; A.nmatches = (int *)b[13].p;
    SGX_LAUNCH(k_search_triangulation, dim3(1), dim3(256), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    std::vector<int> m12((size_t)n1);
    SGX_CHECK_HIP(hipMemcpy(m12.data(), A.match12, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    int n = 0;
    for (int i = 0; i < n1; i++) if (m12[(size_t)i] >= 0) { pairs[2 * n] = i; pairs[2 * n + 1] = m12[(size_t)i]; n++; }      // vMatchedPairs in ascending idx1 (:815-822)
    *npairs = n;
    return SGX_OK;
}

// shared body of the two SearchByBoW overloads (ORBmatcher.cc:167-290 KeyFrame-Frame, :524-655 KeyFrame-KeyFrame)
static int search_by_bow_impl(
    int nk, const sgx_keypoint *keys_kf_un, const uint8_t *desc_kf, const uint8_t *kf_good_mp, const int32_t *feat_node_kf,
    int nf, const sgx_keypoint *keys_f_un, const uint8_t *desc_f, const uint8_t *good_f, const int32_t *feat_node_f, float nnratio, int check_orientation, int th_low,
    int32_t *match_f, int32_t *nmatches)
{
    std::vector<int> it1, id1, st1, it2, id2, st2, job;
    group_by_node(feat_node_kf, nk, it1, id1, st1); group_by_node(feat_node_f, nf, it2, id2, st2);
    for (size_t a = 0, b = 0; a < id1.size() && b < id2.size();) {
        if (id1[a] == id2[b]) { job.push_back(st1[a]); job.push_back(st1[a + 1]); job.push_back(st2[b]); job.push_back(st2[b + 1]); a++; b++; }
        else if (id1[a] < id2[b]) a++; else b++;
    }
    SgxBowArgs A; memset(&A, 0, sizeof A);
    A.nk = nk; A.nf = nf; A.nnodes = (int)(job.size() / 4); A.nnratio = nnratio; A.check_ori = check_orientation; A.th_low = th_low;
    SgxStaged b[12]; int rc; const int dummy = 0;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, keys_kf_un, (size_t)nk * 28); PUT(1, desc_kf, (size_t)nk * 32); PUT(2, kf_good_mp, (size_t)nk);
    PUT(3, keys_f_un, (size_t)nf * 28); PUT(4, desc_f, (size_t)nf * 32);
    PUT(5, it1.empty() ? &dummy : it1.data(), it1.empty() ? 4 : it1.size() * 4); PUT(6, it2.empty() ? &dummy : it2.data(), it2.empty() ? 4 : it2.size() * 4);
    PUT(7, job.empty() ? &dummy : job.data(), job.empty() ? 4 : job.size() * 4);
    PUT(8, nullptr, (size_t)nf * 4); PUT(9, nullptr, 4);
    if (good_f) { PUT(10, good_f, (size_t)nf); A.good_f = (const uint8_t *)b[10].p; }
#undef PUT
    A.keys_k = (const uint8_t *)b[0].p; A.desc_k = (const uint32_t *)b[1].p; A.good_k = (const uint8_t *)b[2].p;
    A.keys_f = (const uint8_t *)b[3].p; A.desc_f = (const uint32_t *)b[4].p;
    A.items_k = (const int *)b[5].p; A.items_f = (const int *)b[6].p; A.job = (const int *)b[7].p; A.match_f = (int *)b[8].p; A.nmatches = (int *)b[9].p;
    SGX_LAUNCH(k_search_bow, dim3(1), dim3(256), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(match_f, A.match_f, (size_t)nf * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(nmatches, A.nmatches, 4, hipMemcpyDeviceToHost));
    return SGX_OK;
}

extern "C" int sgx_match_search_by_bow(
    int nk, const sgx_keypoint *keys_kf_un, const uint8_t *desc_kf, const uint8_t *kf_good_mp, const int32_t *feat_node_kf,
    int nf, const sgx_keypoint *keys_f_un, const uint8_t *desc_f, const int32_t *feat_node_f, float nnratio, int check_orientation,
    int32_t *match_f, int32_t *nmatches)
{
    if (nk < 0 || nf < 0 || !nmatches || (nf > 0 && !match_f)) return SGX_ERR_INVALID;
    *nmatches = 0;
    for (int j = 0; j < nf; j++) match_f[j] = -1;
    if (nk == 0 || nf == 0) return SGX_OK;
    if (!keys_kf_un || !desc_kf || !kf_good_mp || !feat_node_kf || !keys_f_un || !desc_f || !feat_node_f) return SGX_ERR_INVALID;
    return search_by_bow_impl(nk, keys_kf_un, desc_kf, kf_good_mp, feat_node_kf, nf, keys_f_un, desc_f, nullptr, feat_node_f, nnratio, check_orientation, SGX_TH_LOW, match_f, nmatches);
}

extern "C" int sgx_match_search_by_bow_kf(
    int n1, const sgx_keypoint *keys1_un, const uint8_t *desc1, const uint8_t *good1, const int32_t *feat_node1,
    int n2, const sgx_keypoint *keys2_un, const uint8_t *desc2, const uint8_t *good2, const int32_t *feat_node2, float nnratio, int check_orientation,
    int32_t *match12, int32_t *nmatches)
{
    if (n1 < 0 || n2 < 0 || !nmatches || (n1 > 0 && !match12)) return SGX_ERR_INVALID;
    *nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return SGX_OK;
    if (!keys1_un || !desc1 || !good1 || !feat_node1 || !keys2_un || !desc2 || !good2 || !feat_node2) return SGX_ERR_INVALID;
    std::vector<int32_t> m2((size_t)n2, -1);
    const int rc = search_by_bow_impl(n1, keys1_un, desc1, good1, feat_node1, n2, keys2_un, desc2, good2, feat_node2, nnratio, check_orientation, SGX_TH_LOW - 1, m2.data(), nmatches);
    if (rc != SGX_OK) return rc;
    for (int j = 0; j < n2; j++) if (m2[(size_t)j] >= 0) match12[m2[(size_t)j]] = j;     // vbMatched2 makes the relation one-to-one, so the inverse is vpMatches12 (:600)
    return SGX_OK;
}

extern "C" int sgx_match_fuse_search(
    int nk, const sgx_keypoint *keys_un, const uint8_t *desc, const float *uright, const float *Tcw,
    int nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
    const sgx_camera *cam, const float *scale_factors, const float *inv_level_sigma2, int nlevels, float log_scale_factor, float th,
    int32_t *best_idx, int32_t *best_dist, int32_t *nfused)
{
    if (nk < 0 || nm < 0 || !cam || !Tcw || !scale_factors || !inv_level_sigma2 || nlevels < 1 || nlevels > 12 || !nfused || (nm > 0 && (!best_idx || !best_dist))) return SGX_ERR_INVALID;
    *nfused = 0;
    for (int i = 0; i < nm; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (nk == 0 || nm == 0) return SGX_OK;
    if (!keys_un || !desc || !uright || !m_xw || !m_normal || !m_min_dist || !m_max_dist || !m_desc || !m_skip) return SGX_ERR_INVALID;
    if (!octaves_ok(keys_un, nk, nlevels)) return SGX_ERR_INVALID;
    // the keyframe's mGrid (Frame::AssignFeaturesToGrid / PosInGrid: round(), Frame.cc:257-272, :409-419) as CSR, cell (ix, iy) -> ix * 48 + iy, insertion (index) order inside a cell
    const float invW = 64.0f / (cam->max_x - cam->min_x), invH = 48.0f / (cam->max_y - cam->min_y);
    std::vector<int> cell((size_t)nk, -1), start(64 * 48 + 1, 0), items;
    for (int i = 0; i < nk; i++) {
        const int px = (int)round((keys_un[i].x - cam->min_x) * invW), py = (int)round((keys_un[i].y - cam->min_y) * invH);
        if (px < 0 || px >= 64 || py < 0 || py >= 48) continue;
        cell[(size_t)i] = px * 48 + py; start[(size_t)cell[(size_t)i] + 1]++;
    }
    for (int c = 0; c < 64 * 48; c++) start[(size_t)c + 1] += start[(size_t)c];
    items.resize((size_t)start[64 * 48] > 0 ? (size_t)start[64 * 48] : 1);
    { std::vector<int> fill(64 * 48, 0); for (int i = 0; i < nk; i++) if (cell[(size_t)i] >= 0) items[(size_t)start[(size_t)cell[(size_t)i]] + fill[(size_t)cell[(size_t)i]]++] = i; }
    SgxFuseArgs A; memset(&A, 0, sizeof A);
    A.nk = nk; A.nm = nm; A.nlevels = nlevels; A.log_scale_factor = log_scale_factor; A.th = th;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) A.Rcw[r][c] = Tcw[4 * r + c]; A.tcw[r] = Tcw[4 * r + 3]; }
    for (int r = 0; r < 3; r++) A.Ow[r] = -(A.Rcw[0][r] * A.tcw[0] + A.Rcw[1][r] * A.tcw[1] + A.Rcw[2][r] * A.tcw[2]);
    A.cam.fx = cam->fx; A.cam.fy = cam->fy; A.cam.cx = cam->cx; A.cam.cy = cam->cy; A.cam.bf = cam->bf; A.cam.minX = cam->min_x; A.cam.maxX = cam->max_x; A.cam.minY = cam->min_y; A.cam.maxY = cam->max_y;
    for (int i = 0; i < nlevels; i++) { A.scale.s[i] = scale_factors[i]; A.inv_sigma2.s[i] = inv_level_sigma2[i]; }
    SgxStaged b[14]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, keys_un, (size_t)nk * 28); PUT(1, desc, (size_t)nk * 32); PUT(2, uright, (size_t)nk * 4);
    PUT(3, start.data(), start.size() * 4); PUT(4, items.data(), items.size() * 4);
    PUT(5, m_xw, (size_t)nm * 12); PUT(6, m_normal, (size_t)nm * 12); PUT(7, m_min_dist, (size_t)nm * 4); PUT(8, m_max_dist, (size_t)nm * 4); PUT(9, m_desc, (size_t)nm * 32); PUT(10, m_skip, (size_t)nm);
    PUT(11, nullptr, (size_t)nm * 4); PUT(12, nullptr, (size_t)nm * 4);
#undef PUT
    A.keys = (const uint8_t *)b[0].p; A.desc = (const uint32_t *)b[1].p; A.uright = (const float *)b[2].p; A.cell_start = (const int *)b[3].p; A.cell_items = (const int *)b[4].p;
    A.m_xw = (const float *)b[5].p; A.m_normal = (const float *)b[6].p; A.m_min_dist = (const float *)b[7].p; A.m_max_dist = (const float *)b[8].p;
    A.m_desc = (const uint32_t *)b[9].p; A.m_skip = (const uint8_t *)b[10].p; A.best_idx = (int *)b[11].p; A.best_dist = (int *)b[12].p;
    SGX_LAUNCH(k_fuse_search, dim3((nm + 255) / 256), dim3(256), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(best_idx, A.best_idx, (size_t)nm * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(best_dist, A.best_dist, (size_t)nm * 4, hipMemcpyDeviceToHost));
    int n = 0; for (int i = 0; i < nm; i++) n += best_idx[i] >= 0;
    *nfused = n;
    return SGX_OK;
}

extern "C" int sgx_match_project_keyframe(
    int nc, const sgx_keypoint *ckeys_un, const uint8_t *cdesc, const uint8_t *c_has_mp, const float *cTcw,
    int nk, const sgx_keypoint *kf_keys_un, const uint8_t *kf_ok, const float *m_xw, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th, int orb_dist, int check_orientation,
    int32_t *cur_match, int32_t *nmatches)
{
    if (nc < 0 || nk < 0 || !cam || !cTcw || !scale_factors || nlevels < 1 || nlevels > 12 || !nmatches || (nc > 0 && !cur_match)) return SGX_ERR_INVALID;
    *nmatches = 0;
    for (int k = 0; k < nc; k++) cur_match[k] = -1;
    if (nc == 0 || nk == 0) return SGX_OK;
    if (!ckeys_un || !cdesc || !c_has_mp || !kf_keys_un || !kf_ok || !m_xw || !m_min_dist || !m_max_dist || !m_desc) return SGX_ERR_INVALID;
    // CurrentFrame.mGrid (AssignFeaturesToGrid / PosInGrid: round(), Frame.cc:257-272, :409-419) as CSR, cell (ix, iy) -> ix * 48 + iy, index order inside a cell
    const float invW = 64.0f / (cam->max_x - cam->min_x), invH = 48.0f / (cam->max_y - cam->min_y);
    std::vector<int> cell((size_t)nc, -1), start(64 * 48 + 1, 0), items;
    for (int i = 0; i < nc; i++) {
        const int px = (int)round((ckeys_un[i].x - cam->min_x) * invW), py = (int)round((ckeys_un[i].y - cam->min_y) * invH);
        if (px < 0 || px >= 64 || py < 0 || py >= 48) continue;
        cell[(size_t)i] = px * 48 + py; start[(size_t)cell[(size_t)i] + 1]++;
    }
    for (int c = 0; c < 64 * 48; c++) start[(size_t)c + 1] += start[(size_t)c];
    items.resize((size_t)start[64 * 48] > 0 ? (size_t)start[64 * 48] : 1);
    { std::vector<int> fill(64 * 48, 0); for (int i = 0; i < nc; i++) if (cell[(size_t)i] >= 0) items[(size_t)start[(size_t)cell[(size_t)i]] + fill[(size_t)cell[(size_t)i]]++] = i; }
    SgxKfProjArgs A; memset(&A, 0, sizeof A);
    A.nc = nc; A.nk = nk; A.nlevels = nlevels; A.orb_dist = orb_dist; A.check_ori = check_orientation; A.log_scale_factor = log_scale_factor; A.th = th;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) A.Rcw[r][c] = cTcw[4 * r + c]; A.tcw[r] = cTcw[4 * r + 3]; }
    for (int i = 0; i < 3; i++) {                       // Ow = -Rcw.t()*tcw: transpose flag -> generic gemm, double accumulation, alpha = -1
        double s = 0; for (int k = 0; k < 3; k++) s += (double)A.Rcw[k][i] * (double)A.tcw[k];
        A.Ow[i] = (float)(s * -1.0);
    }
    A.cam.fx = cam->fx; A.cam.fy = cam->fy; A.cam.cx = cam->cx; A.cam.cy = cam->cy; A.cam.bf = cam->bf; A.cam.minX = cam->min_x; A.cam.maxX = cam->max_x; A.cam.minY = cam->min_y; A.cam.maxY = cam->max_y;
    for (int i = 0; i < nlevels; i++) A.scale.s[i] = scale_factors[i];
    SgxStaged b[18]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, ckeys_un, (size_t)nc * 28); PUT(1, cdesc, (size_t)nc * 32); PUT(2, c_has_mp, (size_t)nc); PUT(3, start.data(), start.size() * 4); PUT(4, items.data(), items.size() * 4);
    PUT(5, kf_keys_un, (size_t)nk * 28); PUT(6, kf_ok, (size_t)nk); PUT(7, m_xw, (size_t)nk * 12); PUT(8, m_min_dist, (size_t)nk * 4); PUT(9, m_max_dist, (size_t)nk * 4); PUT(10, m_desc, (size_t)nk * 32);
    PUT(11, nullptr, (size_t)nc * 4); PUT(12, nullptr, (size_t)nc * 4); PUT(13, nullptr, (size_t)nk * 4); PUT(14, nullptr, (size_t)nc * 4); PUT(15, nullptr, 4);
#undef PUT
    A.ckeys = (const uint8_t *)b[0].p; A.cdesc = (const uint32_t *)b[1].p; A.c_has_mp = (const uint8_t *)b[2].p; A.cell_start = (const int *)b[3].p; A.cell_items = (const int *)b[4].p;
    A.kf_keys = (const uint8_t *)b[5].p; A.kf_ok = (const uint8_t *)b[6].p; A.m_xw = (const float *)b[7].p; A.m_min_dist = (const float *)b[8].p; A.m_max_dist = (const float *)b[9].p;
    A.m_desc = (const uint32_t *)b[10].p; A.lock_a = (int *)b[11].p; A.lock_b = (int *)b[12].p; A.choice = (int *)b[13].p; A.cur_match = (int *)b[14].p; A.nmatches = (int *)b[15].p;
    SGX_LAUNCH(k_match_project_kf, dim3(1), dim3(1024), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(cur_match, A.cur_match, (size_t)nc * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(nmatches, A.nmatches, 4, hipMemcpyDeviceToHost));
    return SGX_OK;
}

// ---- loop-closing matchers that project through a Sim3 -------------------------------------------------------------------------------------------------------------

// KeyFrame::mGrid (KeyFrame.cc:39-47 copies Frame::mGrid, built by AssignFeaturesToGrid / PosInGrid with round(), Frame.cc:257-272, :409-419) as CSR:
// cell (ix, iy) -> ix * 48 + iy, index order inside a cell
static void build_kf_grid(int n, const sgx_keypoint *keys, const sgx_camera *cam, std::vector<int> &start, std::vector<int> &items)
{
    const float invW = 64.0f / (cam->max_x - cam->min_x), invH = 48.0f / (cam->max_y - cam->min_y);
    std::vector<int> cell((size_t)n, -1);
    start.assign(64 * 48 + 1, 0);
    for (int i = 0; i < n; i++) {
        const int px = (int)round((keys[i].x - cam->min_x) * invW), py = (int)round((keys[i].y - cam->min_y) * invH);
        if (px < 0 || px >= 64 || py < 0 || py >= 48) continue;
        cell[(size_t)i] = px * 48 + py; start[(size_t)cell[(size_t)i] + 1]++;
    }
    for (int c = 0; c < 64 * 48; c++) start[(size_t)c + 1] += start[(size_t)c];
    items.assign((size_t)start[64 * 48] > 0 ? (size_t)start[64 * 48] : 1, 0);
    std::vector<int> fill(64 * 48, 0);
    for (int i = 0; i < n; i++) if (cell[(size_t)i] >= 0) items[(size_t)start[(size_t)cell[(size_t)i]] + fill[(size_t)cell[(size_t)i]]++] = i;
}

// "Decompose Scw" (ORBmatcher.cc:301-306, :990-995).  cv::Mat arithmetic: Mat::dot accumulates in double; Mat / double is a convertTo by the float of 1/scw;
// -Rcw.t() * tcw is a gemm with a transpose flag (double accumulation, alpha = -1).
static void decompose_scw(const float *Scw, float R[3][3], float t[3], float Ow[3])
{
    double s = 0; for (int c = 0; c < 3; c++) s += (double)Scw[c] * (double)Scw[c];
    const float scw = (float)sqrt(s), inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r][c] = Scw[4 * r + c] * inv; t[r] = Scw[4 * r + 3] * inv; }
    for (int i = 0; i < 3; i++) {
        double a = 0; for (int k = 0; k < 3; k++) a += (double)R[k][i] * (double)t[k];
        Ow[i] = (float)(a * -1.0);
    }
}

static void sim3_fill_common(SgxSim3ProjArgs &A, const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th)
{
    A.nlevels = nlevels; A.log_scale_factor = log_scale_factor; A.th = th;
    A.cam.fx = cam->fx; A.cam.fy = cam->fy; A.cam.cx = cam->cx; A.cam.cy = cam->cy; A.cam.bf = cam->bf; A.cam.minX = cam->min_x; A.cam.maxX = cam->max_x; A.cam.minY = cam->min_y; A.cam.maxY = cam->max_y;
    for (int i = 0; i < nlevels; i++) A.scale.s[i] = scale_factors[i];
}

extern "C" int sgx_match_fuse_search_sim3(
    int nk, const sgx_keypoint *keys_un, const uint8_t *desc, const float *Scw,
    int nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float th,
    int32_t *best_idx, int32_t *best_dist, int32_t *nfused)
{
    if (nk < 0 || nm < 0 || !cam || !Scw || !scale_factors || nlevels < 1 || nlevels > 12 || !nfused || (nm > 0 && (!best_idx || !best_dist))) return SGX_ERR_INVALID;
    *nfused = 0;
    for (int i = 0; i < nm; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (nk == 0 || nm == 0) return SGX_OK;
    if (!keys_un || !desc || !m_xw || !m_normal || !m_min_dist || !m_max_dist || !m_desc || !m_skip) return SGX_ERR_INVALID;
    std::vector<int> start, items;
    build_kf_grid(nk, keys_un, cam, start, items);
    SgxSim3ProjArgs A; memset(&A, 0, sizeof A);
    A.nk = nk; A.nm = nm; A.flags = SGX_S3_NORMAL; A.th_accept = SGX_TH_LOW;
    sim3_fill_common(A, cam, scale_factors, nlevels, log_scale_factor, th);
    decompose_scw(Scw, A.R1, A.t1, A.Ow);
    SgxStaged b[12]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, keys_un, (size_t)nk * 28); PUT(1, desc, (size_t)nk * 32); PUT(2, start.data(), start.size() * 4); PUT(3, items.data(), items.size() * 4);
    PUT(4, m_xw, (size_t)nm * 12); PUT(5, m_normal, (size_t)nm * 12); PUT(6, m_min_dist, (size_t)nm * 4); PUT(7, m_max_dist, (size_t)nm * 4); PUT(8, m_desc, (size_t)nm * 32); PUT(9, m_skip, (size_t)nm);
    PUT(10, nullptr, (size_t)nm * 4); PUT(11, nullptr, (size_t)nm * 4);
#undef PUT
    A.keys = (const uint8_t *)b[0].p; A.desc = (const uint32_t *)b[1].p; A.cell_start = (const int *)b[2].p; A.cell_items = (const int *)b[3].p;
    A.m_xw = (const float *)b[4].p; A.m_normal = (const float *)b[5].p; A.m_min_dist = (const float *)b[6].p; A.m_max_dist = (const float *)b[7].p;
    A.m_desc = (const uint32_t *)b[8].p; A.m_skip = (const uint8_t *)b[9].p; A.best_idx = (int *)b[10].p; A.best_dist = (int *)b[11].p;
    SGX_LAUNCH(k_sim3_search, dim3((nm + 255) / 256), dim3(256), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(best_idx, A.best_idx, (size_t)nm * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(best_dist, A.best_dist, (size_t)nm * 4, hipMemcpyDeviceToHost));
    int n = 0; for (int i = 0; i < nm; i++) n += best_idx[i] >= 0;
    *nfused = n;
    return SGX_OK;
}

extern "C" int sgx_match_project_sim3(
    int nk, const sgx_keypoint *keys_un, const uint8_t *desc, const uint8_t *matched_in, const float *Scw,
    int nm, const float *m_xw, const float *m_normal, const float *m_min_dist, const float *m_max_dist, const uint8_t *m_desc, const uint8_t *m_skip,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, int th,
    int32_t *matched_out, int32_t *nmatches)
{
    if (nk < 0 || nm < 0 || !cam || !Scw || !scale_factors || nlevels < 1 || nlevels > 12 || !nmatches || (nk > 0 && !matched_out)) return SGX_ERR_INVALID;
    *nmatches = 0;
    for (int k = 0; k < nk; k++) matched_out[k] = -1;
    if (nk == 0 || nm == 0) return SGX_OK;
    if (!keys_un || !desc || !matched_in || !m_xw || !m_normal || !m_min_dist || !m_max_dist || !m_desc || !m_skip) return SGX_ERR_INVALID;
    std::vector<int> start, items;
    build_kf_grid(nk, keys_un, cam, start, items);
    SgxSim3ProjArgs A; memset(&A, 0, sizeof A);
    A.nk = nk; A.nm = nm; A.flags = SGX_S3_NORMAL | SGX_S3_FLOAT_INVZ; A.th_accept = SGX_TH_LOW;
    sim3_fill_common(A, cam, scale_factors, nlevels, log_scale_factor, (float)th);
    decompose_scw(Scw, A.R1, A.t1, A.Ow);
    SgxStaged b[17]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, keys_un, (size_t)nk * 28); PUT(1, desc, (size_t)nk * 32); PUT(2, start.data(), start.size() * 4); PUT(3, items.data(), items.size() * 4);
    PUT(4, m_xw, (size_t)nm * 12); PUT(5, m_normal, (size_t)nm * 12); PUT(6, m_min_dist, (size_t)nm * 4); PUT(7, m_max_dist, (size_t)nm * 4); PUT(8, m_desc, (size_t)nm * 32); PUT(9, m_skip, (size_t)nm);
    PUT(10, nullptr, (size_t)nm * 4); PUT(11, nullptr, (size_t)nm * 4);
    PUT(12, matched_in, (size_t)nk); PUT(13, nullptr, (size_t)nk * 4); PUT(14, nullptr, (size_t)nk * 4); PUT(15, nullptr, (size_t)nk * 4); PUT(16, nullptr, 4);
#undef PUT
    A.keys = (const uint8_t *)b[0].p; A.desc = (const uint32_t *)b[1].p; A.cell_start = (const int *)b[2].p; A.cell_items = (const int *)b[3].p;
    A.m_xw = (const float *)b[4].p; A.m_normal = (const float *)b[5].p; A.m_min_dist = (const float *)b[6].p; A.m_max_dist = (const float *)b[7].p;
    A.m_desc = (const uint32_t *)b[8].p; A.m_skip = (const uint8_t *)b[9].p; A.best_idx = (int *)b[10].p; A.best_dist = (int *)b[11].p;
    A.taken_in = (const uint8_t *)b[12].p; A.lock_a = (int *)b[13].p; A.lock_b = (int *)b[14].p; A.matched_out = (int *)b[15].p; A.nmatches = (int *)b[16].p;
    SGX_LAUNCH(k_sim3_search_locked, dim3(1), dim3(1024), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(matched_out, A.matched_out, (size_t)nk * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(nmatches, A.nmatches, 4, hipMemcpyDeviceToHost));
    return SGX_OK;
}

extern "C" int sgx_match_search_by_sim3(
    int n1, const sgx_keypoint *keys1_un, const uint8_t *desc1, const float *Tcw1, const uint8_t *mp_ok1, const float *m_xw1, const float *m_min_dist1, const float *m_max_dist1, const uint8_t *m_desc1,
    int n2, const sgx_keypoint *keys2_un, const uint8_t *desc2, const float *Tcw2, const uint8_t *mp_ok2, const float *m_xw2, const float *m_min_dist2, const float *m_max_dist2, const uint8_t *m_desc2,
    const sgx_camera *cam, const float *scale_factors, int nlevels, float log_scale_factor, float s12, const float *R12, const float *t12, float th,
    int32_t *match12, int32_t *nfound)
{
    if (n1 < 0 || n2 < 0 || !cam || !scale_factors || nlevels < 1 || nlevels > 12 || !nfound || !R12 || !t12 || !Tcw1 || !Tcw2 || (n1 > 0 && !match12)) return SGX_ERR_INVALID;
    *nfound = 0;
    if (n1 == 0 || n2 == 0) return SGX_OK;
    if (!keys1_un || !desc1 || !mp_ok1 || !m_xw1 || !m_min_dist1 || !m_max_dist1 || !m_desc1 || !keys2_un || !desc2 || !mp_ok2 || !m_xw2 || !m_min_dist2 || !m_max_dist2 || !m_desc2) return SGX_ERR_INVALID;
    // vbAlreadyMatched1 / vbAlreadyMatched2 (:1136-1147) folded into the per-point skip flags
    std::vector<uint8_t> skip1((size_t)n1), skip2((size_t)n2);
    for (int i = 0; i < n2; i++) skip2[(size_t)i] = !mp_ok2[i];
    for (int i = 0; i < n1; i++) {
        skip1[(size_t)i] = !mp_ok1[i] || match12[i] != -1;
        if (match12[i] >= 0 && match12[i] < n2) skip2[(size_t)match12[i]] = 1;
    }
    // s12*R12, (1.0/s12)*R12.t(), -sR21*t12 (:1124-1126): float scaling by the float of the factor; the 3x3 * 3x1 product goes through cv::gemm's small path
    float sR12[3][3], sR21[3][3], t21[3];
    const float inv_s = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sR12[r][c] = R12[3 * r + c] * s12; sR21[r][c] = R12[3 * c + r] * inv_s; }
    for (int r = 0; r < 3; r++) { const float t = sR21[r][0] * t12[0] + sR21[r][1] * t12[1] + sR21[r][2] * t12[2]; t21[r] = (float)((double)t * -1.0 + 0.0); }
    std::vector<int> start1, items1, start2, items2;
    build_kf_grid(n1, keys1_un, cam, start1, items1); build_kf_grid(n2, keys2_un, cam, start2, items2);
    SgxStaged b[24]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, keys1_un, (size_t)n1 * 28); PUT(1, desc1, (size_t)n1 * 32); PUT(2, start1.data(), start1.size() * 4); PUT(3, items1.data(), items1.size() * 4);
    PUT(4, m_xw1, (size_t)n1 * 12); PUT(5, m_min_dist1, (size_t)n1 * 4); PUT(6, m_max_dist1, (size_t)n1 * 4); PUT(7, m_desc1, (size_t)n1 * 32); PUT(8, skip1.data(), (size_t)n1);
    PUT(9, nullptr, (size_t)n1 * 4); PUT(10, nullptr, (size_t)n1 * 4);
    PUT(11, keys2_un, (size_t)n2 * 28); PUT(12, desc2, (size_t)n2 * 32); PUT(13, start2.data(), start2.size() * 4); PUT(14, items2.data(), items2.size() * 4);
    PUT(15, m_xw2, (size_t)n2 * 12); PUT(16, m_min_dist2, (size_t)n2 * 4); PUT(17, m_max_dist2, (size_t)n2 * 4); PUT(18, m_desc2, (size_t)n2 * 32); PUT(19, skip2.data(), (size_t)n2);
    PUT(20, nullptr, (size_t)n2 * 4); PUT(21, nullptr, (size_t)n2 * 4);
#undef PUT
    SgxSim3ProjArgs A; memset(&A, 0, sizeof A);
    A.flags = SGX_S3_TWO_STEP | SGX_S3_CAM_DIST; A.th_accept = SGX_TH_HIGH;
    sim3_fill_common(A, cam, scale_factors, nlevels, log_scale_factor, th);
    SgxSim3ProjArgs B = A;
    // KF1's points into KF2 (:1150-1224)
    A.nk = n2; A.nm = n1;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { A.R1[r][c] = Tcw1[4 * r + c]; A.R2[r][c] = sR21[r][c]; } A.t1[r] = Tcw1[4 * r + 3]; A.t2[r] = t21[r]; }
    A.keys = (const uint8_t *)b[11].p; A.desc = (const uint32_t *)b[12].p; A.cell_start = (const int *)b[13].p; A.cell_items = (const int *)b[14].p;
    A.m_xw = (const float *)b[4].p; A.m_min_dist = (const float *)b[5].p; A.m_max_dist = (const float *)b[6].p; A.m_desc = (const uint32_t *)b[7].p; A.m_skip = (const uint8_t *)b[8].p;
    A.best_idx = (int *)b[9].p; A.best_dist = (int *)b[10].p;
    SGX_LAUNCH(k_sim3_search, dim3((n1 + 255) / 256), dim3(256), (sgx_stream_t)0, A);
    // KF2's points into KF1 (:1227-1301)
    B.nk = n1; B.nm = n2;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { B.R1[r][c] = Tcw2[4 * r + c]; B.R2[r][c] = sR12[r][c]; } B.t1[r] = Tcw2[4 * r + 3]; B.t2[r] = t12[r]; }
    B.keys = (const uint8_t *)b[0].p; B.desc = (const uint32_t *)b[1].p; B.cell_start = (const int *)b[2].p; B.cell_items = (const int *)b[3].p;
    B.m_xw = (const float *)b[15].p; B.m_min_dist = (const float *)b[16].p; B.m_max_dist = (const float *)b[17].p; B.m_desc = (const uint32_t *)b[18].p; B.m_skip = (const uint8_t *)b[19].p;
    B.best_idx = (int *)b[20].p; B.best_dist = (int *)b[21].p;
    SGX_LAUNCH(k_sim3_search, dim3((n2 + 255) / 256), dim3(256), (sgx_stream_t)0, B);
    SGX_CHECK_HIP(hipGetLastError());
    std::vector<int> m1((size_t)n1), m2((size_t)n2);
    SGX_CHECK_HIP(hipMemcpy(m1.data(), A.best_idx, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(m2.data(), B.best_idx, (size_t)n2 * 4, hipMemcpyDeviceToHost));
    int n = 0;
    for (int i1 = 0; i1 < n1; i1++) {                                   // check agreement (:1305-1320)
        const int idx2 = m1[(size_t)i1];
        if (idx2 >= 0 && m2[(size_t)idx2] == i1) { match12[i1] = idx2; n++; }
    }
    *nfound = n;
    return SGX_OK;
}

extern "C" int sgx_match_search_for_initialization(
    int n1, const sgx_keypoint *keys1_un, const uint8_t *desc1, int n2, const sgx_keypoint *keys2_un, const uint8_t *desc2,
    float *prev_matched, int window_size, float nnratio, int check_orientation, const sgx_camera *cam, int32_t *matches12, int32_t *nmatches)
{
    if (n1 < 0 || n2 < 0 || !cam || !nmatches || window_size < 0 || (n1 > 0 && (!matches12 || !prev_matched))) return SGX_ERR_INVALID;
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return SGX_OK;
    if (!keys1_un || !desc1 || !keys2_un || !desc2 || n2 >= (1 << 22)) return SGX_ERR_INVALID;
    std::vector<int> start, items;
    build_kf_grid(n2, keys2_un, cam, start, items);              // F2.mGrid (Frame::AssignFeaturesToGrid, same round() cell rule)
    SgxInitSearchArgs A; memset(&A, 0, sizeof A);
    A.n1 = n1; A.n2 = n2; A.window = window_size; A.check_ori = check_orientation; A.nnratio = nnratio;
    A.cam.fx = cam->fx; A.cam.fy = cam->fy; A.cam.cx = cam->cx; A.cam.cy = cam->cy; A.cam.bf = cam->bf; A.cam.minX = cam->min_x; A.cam.maxX = cam->max_x; A.cam.minY = cam->min_y; A.cam.maxY = cam->max_y;
    SgxStaged b[12]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, keys1_un, (size_t)n1 * 28); PUT(1, desc1, (size_t)n1 * 32); PUT(2, keys2_un, (size_t)n2 * 28); PUT(3, desc2, (size_t)n2 * 32);
    PUT(4, start.data(), start.size() * 4); PUT(5, items.data(), items.size() * 4); PUT(6, prev_matched, (size_t)n1 * 8);
    PUT(7, nullptr, (size_t)n1 * 4); PUT(8, nullptr, (size_t)n2 * 4); PUT(9, nullptr, (size_t)n2 * 4); PUT(10, nullptr, 4);
#undef PUT
    A.keys1 = (const uint8_t *)b[0].p; A.desc1 = (const uint32_t *)b[1].p; A.keys2 = (const uint8_t *)b[2].p; A.desc2 = (const uint32_t *)b[3].p;
    A.cell_start = (const int *)b[4].p; A.cell_items = (const int *)b[5].p; A.prev_matched = (float *)b[6].p;
    A.matches12 = (int *)b[7].p; A.matches21 = (int *)b[8].p; A.dist21 = (int *)b[9].p; A.nmatches = (int *)b[10].p;
    SGX_LAUNCH(k_search_initialization, dim3(1), dim3(256), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(matches12, A.matches12, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(prev_matched, A.prev_matched, (size_t)n1 * 8, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(nmatches, A.nmatches, 4, hipMemcpyDeviceToHost));
    return SGX_OK;
}

// ---- MapPoint post-steps (MapPoint.cc:242-307, :330-371), batched over points; host pointers, synchronous ------------------------------------------------------------
extern "C" int sgx_mappoint_update_normal_and_depth(int n, const float *xw, const int32_t *obs_start, const float *obs_center, const float *ref_center, const int32_t *ref_level,
                                                    const float *scale_factors, int nlevels, float *normal, float *min_dist, float *max_dist)
{
    if (n < 0 || nlevels < 1 || nlevels > 12 || !scale_factors || (n > 0 && (!xw || !obs_start || !ref_center || !ref_level || !normal || !min_dist || !max_dist))) return SGX_ERR_INVALID;
    if (n == 0) return SGX_OK;
    const int total = obs_start[n];
    if (total < 0 || (total > 0 && !obs_center)) return SGX_ERR_INVALID;
    for (int p = 0; p < n; p++) if (obs_start[p] > obs_start[p + 1] || ref_level[p] < 0 || ref_level[p] >= nlevels) return SGX_ERR_INVALID;
    SgxScales sc; memset(&sc, 0, sizeof sc); for (int i = 0; i < nlevels; i++) sc.s[i] = scale_factors[i];
    SgxStaged b[8]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, xw, (size_t)n * 12); PUT(1, obs_start, ((size_t)n + 1) * 4); PUT(2, obs_center, (size_t)total * 12); PUT(3, ref_center, (size_t)n * 12); PUT(4, ref_level, (size_t)n * 4);
    PUT(5, normal, (size_t)n * 12); PUT(6, min_dist, (size_t)n * 4); PUT(7, max_dist, (size_t)n * 4);          // points without observations keep the caller's values
#undef PUT
    SGX_LAUNCH(k_mappoint_normal_depth, dim3((n + 255) / 256), dim3(256), (sgx_stream_t)0, n, (const float *)b[0].p, (const int *)b[1].p, (const float *)b[2].p, (const float *)b[3].p,
               (const int *)b[4].p, sc, nlevels, (float *)b[5].p, (float *)b[6].p, (float *)b[7].p);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(normal, b[5].p, (size_t)n * 12, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(min_dist, b[6].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(max_dist, b[7].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SGX_OK;
}

extern "C" int sgx_mappoint_distinctive_descriptors(int n, const int32_t *obs_start, const uint8_t *obs_desc, int32_t *best, uint8_t *desc_out)
{
    if (n < 0 || (n > 0 && (!obs_start || !best))) return SGX_ERR_INVALID;
    if (n == 0) return SGX_OK;
    const int total = obs_start[n];
    if (total < 0 || (total > 0 && !obs_desc)) return SGX_ERR_INVALID;
    for (int p = 0; p < n; p++) if (obs_start[p] > obs_start[p + 1] || obs_start[p + 1] - obs_start[p] > 65535) return SGX_ERR_INVALID;
    SgxStaged b[3]; int rc;
    if ((rc = b[0].put(0, obs_start, ((size_t)n + 1) * 4)) != SGX_OK || (rc = b[1].put(1, obs_desc, (size_t)total * 32)) != SGX_OK || (rc = b[2].put(2, nullptr, (size_t)n * 4)) != SGX_OK) return rc;
    SGX_LAUNCH(k_mappoint_distinctive, dim3(n), dim3(64), (sgx_stream_t)0, n, (const int *)b[0].p, (const uint32_t *)b[1].p, (int *)b[2].p);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(best, b[2].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (desc_out) for (int p = 0; p < n; p++) if (best[p] >= 0) memcpy(desc_out + 32 * (size_t)p, obs_desc + 32 * (size_t)(obs_start[p] + best[p]), 32);      // mDescriptor = vDescriptors[BestIdx].clone()
    return SGX_OK;
}

extern "C" int sgx_triangulate_new_map_points(
    int npairs, const int32_t *pairs,
    int n1, const sgx_keypoint *keys1_un, const sgx_keypoint *keys1, const float *uright1, const float *depth1, const float *Tcw1,
    int n2, const sgx_keypoint *keys2_un, const sgx_keypoint *keys2, const float *uright2, const float *depth2, const float *Tcw2,
    const sgx_camera *cam, const float *scale_factors, const float *level_sigma2, int nlevels, uint8_t *ok, float *x3d, int32_t *nnew)
{
    if (npairs < 0 || n1 < 0 || n2 < 0 || !cam || !scale_factors || !level_sigma2 || nlevels < 2 || nlevels > 12 || !Tcw1 || !Tcw2 || !nnew || (npairs > 0 && (!pairs || !ok || !x3d))) return SGX_ERR_INVALID;
    *nnew = 0;
    if (npairs == 0) return SGX_OK;
    if (!keys1_un || !keys1 || !uright1 || !depth1 || !keys2_un || !keys2 || !uright2 || !depth2) return SGX_ERR_INVALID;
    if (!octaves_ok(keys1_un, n1, nlevels) || !octaves_ok(keys2_un, n2, nlevels)) return SGX_ERR_INVALID;
    for (int q = 0; q < npairs; q++) if (pairs[2 * q] < 0 || pairs[2 * q] >= n1 || pairs[2 * q + 1] < 0 || pairs[2 * q + 1] >= n2) return SGX_ERR_INVALID;
    SgxNewPointArgs A; memset(&A, 0, sizeof A);
    A.npairs = npairs; memcpy(A.Tcw1, Tcw1, 64); memcpy(A.Tcw2, Tcw2, 64);
    for (int i = 0; i < 3; i++) {                                 // Ow = -Rcw.t() * tcw (KeyFrame::SetPose, KeyFrame.cc:64-66): transpose flag -> double accumulation, alpha = -1
        double a = 0, b = 0; for (int k = 0; k < 3; k++) { a += (double)Tcw1[4 * k + i] * (double)Tcw1[4 * k + 3]; b += (double)Tcw2[4 * k + i] * (double)Tcw2[4 * k + 3]; }
        A.Ow1[i] = (float)(a * -1.0); A.Ow2[i] = (float)(b * -1.0);
    }
    A.fx = cam->fx; A.fy = cam->fy; A.cx = cam->cx; A.cy = cam->cy; A.mbf = cam->bf; A.ratio_factor = 1.5f * scale_factors[1];       // ratioFactor = 1.5f * mfScaleFactor (:233)
    for (int i = 0; i < nlevels; i++) { A.scale.s[i] = scale_factors[i]; A.sigma2.s[i] = level_sigma2[i]; }
    SgxStaged b[12]; int rc;
#define PUT(k, src, bytes) if ((rc = b[k].put(k, src, bytes)) != SGX_OK) return rc
    PUT(0, pairs, (size_t)npairs * 8); PUT(1, keys1_un, (size_t)n1 * 28); PUT(2, keys1, (size_t)n1 * 28); PUT(3, uright1, (size_t)n1 * 4); PUT(4, depth1, (size_t)n1 * 4);
    PUT(5, keys2_un, (size_t)n2 * 28); PUT(6, keys2, (size_t)n2 * 28); PUT(7, uright2, (size_t)n2 * 4); PUT(8, depth2, (size_t)n2 * 4);
    PUT(9, nullptr, (size_t)npairs); PUT(10, nullptr, (size_t)npairs * 12);
#undef PUT
    A.pairs = (const int *)b[0].p; A.keys1_un = (const uint8_t *)b[1].p; A.keys1 = (const uint8_t *)b[2].p; A.ur1 = (const float *)b[3].p; A.dp1 = (const float *)b[4].p;
    A.keys2_un = (const uint8_t *)b[5].p; A.keys2 = (const uint8_t *)b[6].p; A.ur2 = (const float *)b[7].p; A.dp2 = (const float *)b[8].p; A.ok = (uint8_t *)b[9].p; A.x3d = (float *)b[10].p;
    SGX_LAUNCH(k_triangulate_pairs, dim3((npairs + 255) / 256), dim3(256), (sgx_stream_t)0, A);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(ok, A.ok, (size_t)npairs, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(x3d, A.x3d, (size_t)npairs * 12, hipMemcpyDeviceToHost));
    int n = 0; for (int q = 0; q < npairs; q++) n += ok[q];
    *nnew = n;
    return SGX_OK;
}
