// sgx_match2.cpp — host side of the LocalMapping matcher gates (include/sgx.h): sgx_hamming_matrix, sgx_match_search_for_triangulation.
// Reference behaviour: src/sg-slam/src/ORBmatcher.cc:659-827, :1649-1665.
#include "sgx_match2_kernels.h"
#include "sgx_stage.h"
#include "../../include/sgx.h"
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
