// sgx_voc.cpp — ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) behind the C ABI: loaders, the per-feature descent on the device (sgx_voc_kernels.h),
// BowVector / FeatureVector assembly and L1 scoring on the host.  Reference: src/sg-slam/Thirdparty/DBoW2/DBoW2/{TemplatedVocabulary.h, BowVector.cpp, FeatureVector.cpp,
// ScoringObject.cpp, FORB.cpp}; callers Frame::ComputeBoW (Frame.cc:422-429), KeyFrame::ComputeBoW (KeyFrame.cc:60-69), System::System (System.cc:65-80).
#include "sgx_voc_kernels.h"
#include "sgx_stage.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

struct sgx_voc {
    int k = 0, L = 0, scoring = 0, weighting = 0, nnodes = 0, nwords = 0;
    bool empty = true;
    SgxVocDev dev{};
    std::vector<void *> allocs;
    ~sgx_voc() { for (void *p : allocs) (void)hipFree(p); }
    template <class T> int upload(const std::vector<T> &h, const T **out)
    {
        void *p = nullptr;
        if (hipMalloc(&p, (h.size() ? h.size() : 1) * sizeof(T)) != hipSuccess) return SGX_ERR_NOMEM;
        allocs.push_back(p);
        if (!h.empty() && hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return SGX_ERR_DEVICE;
        *out = (const T *)p; return SGX_OK;
    }
};

extern "C" int sgx_voc_create(int k, int L, int scoring, int weighting, int nnodes, const int32_t *parent, const uint8_t *desc32, const double *weight, const uint8_t *is_leaf, sgx_voc **out)
{
    if (!out || k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3 || nnodes < 1 || (nnodes > 1 && (!parent || !desc32 || !weight || !is_leaf)))
        return SGX_ERR_INVALID;                                   // the header checks of loadFromTextFile (TemplatedVocabulary.h:1373-1377)
    for (int i = 1; i < nnodes; i++) if (parent[i] < 0 || parent[i] >= i) return SGX_ERR_INVALID;      // a node follows its parent in both file formats
    sgx_voc *v = new sgx_voc;
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->nnodes = nnodes;
    std::vector<int> cstart((size_t)nnodes + 1, 0), cidx((size_t)(nnodes > 1 ? nnodes - 1 : 1), 0), word((size_t)nnodes, -1), fill((size_t)nnodes, 0);
    std::vector<uint32_t> d((size_t)nnodes * 8, 0u); std::vector<double> w((size_t)nnodes, 0.0);
    for (int i = 1; i < nnodes; i++) cstart[(size_t)parent[i] + 1]++;
    for (int i = 0; i < nnodes; i++) cstart[(size_t)i + 1] += cstart[(size_t)i];
    for (int i = 1; i < nnodes; i++) {                            // m_nodes[pid].children.push_back(nid) in node order; word ids in node order (:1395-1432)
        const int p = parent[i];
        cidx[(size_t)cstart[(size_t)p] + fill[(size_t)p]++] = i;
        memcpy(&d[(size_t)i * 8], desc32 + (size_t)i * 32, 32); w[(size_t)i] = weight[i];
        if (is_leaf[i]) word[(size_t)i] = v->nwords++;
    }
    v->empty = v->nwords == 0 || cstart[1] == 0;
    int rc;
    if ((rc = v->upload(cstart, &v->dev.child_start)) != SGX_OK || (rc = v->upload(cidx, &v->dev.child_idx)) != SGX_OK || (rc = v->upload(word, &v->dev.word_id)) != SGX_OK ||
        (rc = v->upload(d, &v->dev.desc)) != SGX_OK || (rc = v->upload(w, &v->dev.weight)) != SGX_OK) { delete v; return rc; }
    v->dev.L = L; v->dev.nnodes = nnodes;
    *out = v;
    return SGX_OK;
}

// ".txt" -> loadFromTextFile (TemplatedVocabulary.h:1351-1438), anything else -> loadFromBinaryFile (:1467-1510), as System.cc:69-73 decides
extern "C" int sgx_voc_load(const char *path, sgx_voc **out)
{
    if (!path || !out) return SGX_ERR_INVALID;
    const std::string s(path);
    const bool text = s.size() >= 4 && s.compare(s.size() - 4, 4, ".txt") == 0;
    FILE *f = fopen(path, text ? "r" : "rb");
    if (!f) return SGX_ERR_INVALID;
    int k = 0, L = 0, sc = 0, wg = 0;
    std::vector<int32_t> parent(1, 0); std::vector<uint8_t> desc(32, 0), leaf(1, 0); std::vector<double> weight(1, 0.0);
    bool ok = true;
    if (text) {
        if (fscanf(f, "%d %d %d %d", &k, &L, &sc, &wg) != 4) ok = false;
        while (ok) {
            int pid, isl;
            if (fscanf(f, "%d %d", &pid, &isl) != 2) break;          // end of file (a trailing blank line adds no node here; the reference's eof() loop would append an uninitialised one)
            parent.push_back(pid); leaf.push_back(isl > 0 ? 1 : 0);
            for (int i = 0; i < 32; i++) { int b = 0; if (fscanf(f, "%d", &b) != 1) b = 0; desc.push_back((uint8_t)b); }
            double w = 0; if (fscanf(f, "%lf", &w) != 1) w = 0;
            weight.push_back(w);
        }
    } else {
        unsigned nb = 0, sz = 0;
        if (fread(&nb, 4, 1, f) != 1 || fread(&sz, 4, 1, f) != 1 || fread(&k, 4, 1, f) != 1 || fread(&L, 4, 1, f) != 1 || fread(&sc, 4, 1, f) != 1 || fread(&wg, 4, 1, f) != 1 || sz < 41 || sz > 256) ok = false;
        uint8_t buf[256];
        for (unsigned i = 0; ok && i < nb; i++) {
            if (fread(buf, sz, 1, f) != 1) break;
            int32_t p; float wf; memcpy(&p, buf, 4); memcpy(&wf, buf + 36, 4);
            parent.push_back(p); desc.insert(desc.end(), buf + 4, buf + 36); weight.push_back((double)wf); leaf.push_back(buf[40] != 0);
        }
    }
    fclose(f);
    if (!ok) return SGX_ERR_INVALID;
    return sgx_voc_create(k, L, sc, wg, (int)parent.size(), parent.data(), desc.data(), weight.data(), leaf.data(), out);
}

extern "C" int sgx_voc_info(const sgx_voc *v, int32_t *k, int32_t *L, int32_t *scoring, int32_t *weighting, int32_t *nnodes, int32_t *nwords)
{
    if (!v) return SGX_ERR_INVALID;
    if (k) *k = v->k; if (L) *L = v->L; if (scoring) *scoring = v->scoring; if (weighting) *weighting = v->weighting; if (nnodes) *nnodes = v->nnodes; if (nwords) *nwords = v->nwords;
    return SGX_OK;
}

extern "C" void sgx_voc_destroy(sgx_voc *v) { delete v; }

extern "C" int sgx_voc_transform_batch_dev(sgx_voc *v, const uint8_t *d_desc, size_t desc_pitch, const int32_t *d_n, int batch, int cap, int levelsup,
                                           int32_t *d_word_id, double *d_weight, int32_t *d_feat_node, void *stream)
{
    if (!v || batch < 0 || cap < 0 || (batch > 0 && cap > 0 && (!d_desc || !d_n || !d_word_id || !d_weight || !d_feat_node))) return SGX_ERR_INVALID;
    if (batch == 0 || cap == 0) return SGX_OK;
    if (v->empty) return SGX_ERR_INVALID;                         // transform() on an empty vocabulary returns empty vectors; there is nothing to put into per-feature arrays
    SGX_LAUNCH(k_voc_transform, dim3((cap + 255) / 256, batch), dim3(256), (sgx_stream_t)stream, v->dev, levelsup, d_desc, desc_pitch, d_n, 0, cap, d_word_id, d_weight, d_feat_node);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

static bool must_normalize(int scoring, bool *l2)
{   // ScoringObject.h:73-92: every scoring class is declared with mustNormalize = true (KLScoring too: __SCORING_CLASS(KLScoring, true, L1)) except DotProductScoring
    *l2 = scoring == 1;
    return scoring != 5;
}

extern "C" int sgx_voc_transform(sgx_voc *v, int n, const uint8_t *desc, int levelsup, int32_t *bow_ids, double *bow_weights, int32_t *nbow, int32_t *feat_node, int32_t *feat_word)
{
    if (!v || n < 0 || !nbow || (n > 0 && (!desc || !bow_ids || !bow_weights || !feat_node))) return SGX_ERR_INVALID;
    *nbow = 0;
    for (int i = 0; i < n; i++) { feat_node[i] = -1; if (feat_word) feat_word[i] = -1; }
    if (n == 0 || v->empty) return SGX_OK;                        // :1146-1149
    SgxStaged b[4]; int rc;
    if ((rc = b[0].put(0, desc, (size_t)n * 32)) != SGX_OK || (rc = b[1].put(1, nullptr, (size_t)n * 4)) != SGX_OK || (rc = b[2].put(2, nullptr, (size_t)n * 8)) != SGX_OK ||
        (rc = b[3].put(3, nullptr, (size_t)n * 4)) != SGX_OK) return rc;
    SGX_LAUNCH(k_voc_transform, dim3((n + 255) / 256, 1), dim3(256), (sgx_stream_t)0, v->dev, levelsup, (const uint8_t *)b[0].p, (size_t)0, (const int *)nullptr, n, n,
               (int *)b[1].p, (double *)b[2].p, (int *)b[3].p);
    SGX_CHECK_HIP(hipGetLastError());
    std::vector<int32_t> word((size_t)n); std::vector<double> w((size_t)n);
    SGX_CHECK_HIP(hipMemcpy(word.data(), b[1].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(w.data(), b[2].p, (size_t)n * 8, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(feat_node, b[3].p, (size_t)n * 4, hipMemcpyDeviceToHost));
    // BowVector: std::map keyed by word id, weights accumulated in feature order (addWeight, BowVector.cpp:33-45) or first-wins (addIfNotExist :49-57); FeatureVector = feat_node
    std::map<int32_t, double> bow;
    const bool tf = v->weighting == 0 || v->weighting == 1;       // TF_IDF, TF (:1159) against IDF, BINARY (:1186)
    for (int i = 0; i < n; i++) {
        if (feat_word) feat_word[i] = word[(size_t)i];
        if (!(w[(size_t)i] > 0)) continue;                        // stopped word (:1170)
        auto it = bow.lower_bound(word[(size_t)i]);
        if (it != bow.end() && it->first == word[(size_t)i]) { if (tf) it->second += w[(size_t)i]; }
        else bow.insert(it, std::make_pair(word[(size_t)i], w[(size_t)i]));
    }
    bool l2; const bool must = must_normalize(v->scoring, &l2);
    if (tf && !bow.empty() && !must) { const double nd = (double)bow.size(); for (auto &e : bow) e.second /= nd; }      // :1177-1183
    if (must) {                                                   // BowVector::normalize, BowVector.cpp:61-86
        double norm = 0.0;
        if (!l2) for (auto &e : bow) norm += fabs(e.second);
        else { for (auto &e : bow) norm += e.second * e.second; norm = sqrt(norm); }
        if (norm > 0.0) for (auto &e : bow) e.second /= norm;
    }
    int m = 0;
    for (auto &e : bow) { bow_ids[m] = e.first; bow_weights[m] = e.second; m++; }
    *nbow = m;
    return SGX_OK;
}

// L1Scoring::score (ScoringObject.cpp:23-67) on two BowVectors given as ascending (id, weight) arrays — the scoring the ORB vocabulary files select
extern "C" int sgx_voc_score(const sgx_voc *v, int n1, const int32_t *ids1, const double *w1, int n2, const int32_t *ids2, const double *w2, double *score)
{
    if (!v || !score || n1 < 0 || n2 < 0 || (n1 > 0 && (!ids1 || !w1)) || (n2 > 0 && (!ids2 || !w2))) return SGX_ERR_INVALID;
    if (v->scoring != 0) return SGX_ERR_UNSUPPORTED;
    int a = 0, b = 0; double s = 0;
    while (a < n1 && b < n2) {
        if (ids1[a] == ids2[b]) { s += fabs(w1[a] - w2[b]) - fabs(w1[a]) - fabs(w2[b]); a++; b++; }
        else if (ids1[a] < ids2[b]) { while (a < n1 && ids1[a] < ids2[b]) a++; }
        else { while (b < n2 && ids2[b] < ids1[a]) b++; }
    }
    *score = -s / 2.0;
    return SGX_OK;
}
