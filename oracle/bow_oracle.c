/* bow_oracle.c — TEST INFRASTRUCTURE ONLY (never linked, imported or executed by the product path).
 *
 * CPU restatement of the DBoW2 vocabulary transform the reference runs in Frame::ComputeBoW / KeyFrame::ComputeBoW
 * (src/sg-slam/src/Frame.cc:422-429: mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)).  The DBoW2 sources ARE in the reference tree
 * (src/sg-slam/Thirdparty/DBoW2/DBoW2), so every function cites them directly; what is absent is the vocabulary FILE (ORBvoc.txt / .bin), so the tests run
 * on synthetic vocabularies written in the two file formats the loaders below read.  Parity status: unpinned (no golden vector of the reference's own exists
 * for this path; the reference cannot be built here — OpenCV is absent).
 *
 *   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)      TemplatedVocabulary.h:1231-1273
 *   TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup)  TemplatedVocabulary.h:1139-1206
 *   FORB::distance                                                                FORB.cpp:81-101
 *   BowVector::addWeight / addIfNotExist / normalize                              BowVector.cpp:33-86
 *   FeatureVector::addFeature                                                     FeatureVector.cpp:31-45
 *   L1Scoring::score                                                              ScoringObject.cpp:23-67
 *   loadFromTextFile / loadFromBinaryFile                                         TemplatedVocabulary.h:1351-1438, 1467-1510
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int k, L, scoring, weighting, nnodes, nwords;
    int *parent, *word_id, *child_start, *child_idx;      /* children of node i: child_idx[child_start[i] .. child_start[i + 1]) in insertion (= node id) order */
    uint8_t *desc;                                         /* nnodes x 32 */
    double *weight;
} orc_voc;

/* m_nodes as flat arrays: node 0 is the root; parent[i] (i >= 1) as read from the file; is_leaf[i] > 0 gives the node a word id, in node order (the loaders' rule) */
orc_voc *orc_voc_create(int k, int L, int scoring, int weighting, int nnodes, const int *parent, const uint8_t *desc, const double *weight, const uint8_t *is_leaf)
{
    orc_voc *v = (orc_voc *)calloc(1, sizeof(orc_voc));
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->nnodes = nnodes;
    v->parent = (int *)calloc((size_t)nnodes, sizeof(int)); v->word_id = (int *)malloc(sizeof(int) * (size_t)nnodes);
    v->child_start = (int *)calloc((size_t)nnodes + 1, sizeof(int)); v->child_idx = (int *)malloc(sizeof(int) * (size_t)(nnodes > 0 ? nnodes : 1));
    v->desc = (uint8_t *)calloc((size_t)nnodes, 32); v->weight = (double *)calloc((size_t)nnodes, sizeof(double));
    for (int i = 0; i < nnodes; i++) v->word_id[i] = -1;
    for (int i = 1; i < nnodes; i++) { v->parent[i] = parent[i]; memcpy(v->desc + 32 * (size_t)i, desc + 32 * (size_t)i, 32); v->weight[i] = weight[i]; v->child_start[parent[i] + 1]++; }
    for (int i = 0; i < nnodes; i++) v->child_start[i + 1] += v->child_start[i];
    int *fill = (int *)calloc((size_t)nnodes, sizeof(int));
    for (int i = 1; i < nnodes; i++) { const int p = parent[i]; v->child_idx[v->child_start[p] + fill[p]++] = i; }          /* m_nodes[pid].children.push_back(nid), nid ascending */
    free(fill);
    for (int i = 1; i < nnodes; i++) if (is_leaf[i]) v->word_id[i] = v->nwords++;                                              /* wid = m_words.size() */
    return v;
}
void orc_voc_destroy(orc_voc *v) { if (!v) return; free(v->parent); free(v->word_id); free(v->child_start); free(v->child_idx); free(v->desc); free(v->weight); free(v); }
void orc_voc_info(const orc_voc *v, int *out) { out[0] = v->k; out[1] = v->L; out[2] = v->scoring; out[3] = v->weighting; out[4] = v->nnodes; out[5] = v->nwords; }

/* FORB::distance, FORB.cpp:81-101 */
static int forb_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y; memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24);
    }
    return dist;
}

/* transform(feature, word_id, weight, nid, levelsup), TemplatedVocabulary.h:1231-1273.  A descent that reaches a leaf above level L - levelsup leaves *nid unassigned in
 * the reference (the caller's variable is uninitialised): this restatement reports the leaf itself in that case (and says so). */
static void transform_one(const orc_voc *v, const uint8_t *f, int levelsup, int *word_id, double *weight, int *nid)
{
    const int nid_level = v->L - levelsup;
    int nid_set = 0;
    if (nid_level <= 0) { *nid = 0; nid_set = 1; }
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int s = v->child_start[final_id], e = v->child_start[final_id + 1];
        final_id = v->child_idx[s];
        double best_d = (double)forb_distance(f, v->desc + 32 * (size_t)final_id);
        for (int c = s + 1; c < e; c++) {
            const int id = v->child_idx[c];
            const double d = (double)forb_distance(f, v->desc + 32 * (size_t)id);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (current_level == nid_level) { *nid = final_id; nid_set = 1; }
    } while (v->child_start[final_id + 1] > v->child_start[final_id]);       /* !isLeaf() */
    if (!nid_set) *nid = final_id;
    *word_id = v->word_id[final_id];
    *weight = v->weight[final_id];
}

/* per-feature part of transform(features, v, fv, levelsup): word, weight, node; feat_node = -1 for a stopped word (w <= 0: not added to either vector, :1170, :1198) */
void orc_voc_transform_features(const orc_voc *v, int n, const uint8_t *desc, int levelsup, int *word_id, double *weight, int *feat_node)
{
    for (int i = 0; i < n; i++) {
        word_id[i] = -1; weight[i] = 0; feat_node[i] = -1;
        if (v->nwords == 0 || v->child_start[1] == 0) continue;              /* empty() */
        int w, nd; double wt;
        transform_one(v, desc + 32 * (size_t)i, levelsup, &w, &wt, &nd);
        word_id[i] = w; weight[i] = wt; feat_node[i] = wt > 0 ? nd : -1;
    }
}

static int must_normalize(int scoring, int *l2)
{   /* ScoringObject.h:73-89: L1 (true, L1), L2 (true, L2), ChiSquare (true, L1), KL (true, L1), Bhattacharyya (true, L1), DotProduct (false, L1) */
    *l2 = scoring == 1;
    return scoring != 5;
}

/* the BowVector of transform(features, v, fv, levelsup): ids ascending (std::map order), weights accumulated in feature order, then normalised.  Returns its size. */
int orc_voc_bow_vector(const orc_voc *v, int n, const int *word_id, const double *weight, int *ids, double *w)
{
    int m = 0;
    const int tf = v->weighting == 0 || v->weighting == 1;                   /* TF_IDF = 0, TF = 1, IDF = 2, BINARY = 3 (BowVector.h) */
    for (int i = 0; i < n; i++) {
        if (!(weight[i] > 0)) continue;
        int lo = 0, hi = m;                                                  /* lower_bound */
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (ids[mid] < word_id[i]) lo = mid + 1; else hi = mid; }
        if (lo < m && ids[lo] == word_id[i]) { if (tf) w[lo] += weight[i]; }      /* addWeight :33-45 / addIfNotExist :49-57 */
        else { memmove(ids + lo + 1, ids + lo, sizeof(int) * (size_t)(m - lo)); memmove(w + lo + 1, w + lo, sizeof(double) * (size_t)(m - lo)); ids[lo] = word_id[i]; w[lo] = weight[i]; m++; }
    }
    int l2; const int must = must_normalize(v->scoring, &l2);
    if (tf && m > 0 && !must) { const double nd = (double)m; for (int i = 0; i < m; i++) w[i] /= nd; }      /* :1177-1183 */
    if (must) {                                                              /* BowVector::normalize :61-86 */
        double norm = 0.0;
        if (!l2) for (int i = 0; i < m; i++) norm += fabs(w[i]);
        else { for (int i = 0; i < m; i++) norm += w[i] * w[i]; norm = sqrt(norm); }
        if (norm > 0.0) for (int i = 0; i < m; i++) w[i] /= norm;
    }
    return m;
}

/* L1Scoring::score, ScoringObject.cpp:23-67 */
double orc_bow_score_l1(int n1, const int *id1, const double *w1, int n2, const int *id2, const double *w2)
{
    int a = 0, b = 0; double score = 0;
    while (a < n1 && b < n2) {
        if (id1[a] == id2[b]) { score += fabs(w1[a] - w2[b]) - fabs(w1[a]) - fabs(w2[b]); a++; b++; }
        else if (id1[a] < id2[b]) { while (a < n1 && id1[a] < id2[b]) a++; }          /* lower_bound */
        else { while (b < n2 && id2[b] < id1[a]) b++; }
    }
    return -score / 2.0;
}

/* loadFromTextFile :1351-1438 ("k L scoring weighting", then one line per node: parent is_leaf 32 bytes weight) and loadFromBinaryFile :1467-1510
 * (nb_nodes, size_node, k, L, scoring, weighting, then records of size_node bytes: int parent, 32 bytes, float weight, uchar is_leaf).  System.cc:69-73 picks by the .txt suffix. */
orc_voc *orc_voc_load(const char *path)
{
    const size_t pl = strlen(path);
    const int text = pl >= 4 && strcmp(path + pl - 4, ".txt") == 0;
    FILE *f = fopen(path, text ? "r" : "rb");
    if (!f) return NULL;
    int k = 0, L = 0, sc = 0, wg = 0, cap = 1024, n = 1;
    int *parent = (int *)malloc(sizeof(int) * cap); uint8_t *desc = (uint8_t *)malloc(32 * (size_t)cap); double *weight = (double *)malloc(sizeof(double) * cap); uint8_t *leaf = (uint8_t *)malloc(cap);
    parent[0] = 0; memset(desc, 0, 32); weight[0] = 0; leaf[0] = 0;
#define GROW() if (n == cap) { cap *= 2; parent = (int *)realloc(parent, sizeof(int) * cap); desc = (uint8_t *)realloc(desc, 32 * (size_t)cap); weight = (double *)realloc(weight, sizeof(double) * cap); leaf = (uint8_t *)realloc(leaf, cap); }
    if (text) {
        if (fscanf(f, "%d %d %d %d", &k, &L, &sc, &wg) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || sc < 0 || sc > 5 || wg < 0 || wg > 3) { fclose(f); free(parent); free(desc); free(weight); free(leaf); return NULL; }
        for (;;) {
            int pid, isl;
            if (fscanf(f, "%d %d", &pid, &isl) != 2) break;
            GROW();
            parent[n] = pid; leaf[n] = isl > 0;
            for (int i = 0; i < 32; i++) { int b = 0; if (fscanf(f, "%d", &b) != 1) b = 0; desc[32 * (size_t)n + i] = (uint8_t)b; }
            double w = 0; if (fscanf(f, "%lf", &w) != 1) w = 0;
            weight[n] = w; n++;
        }
    } else {
        unsigned nb = 0, sz = 0;
        if (fread(&nb, 4, 1, f) != 1 || fread(&sz, 4, 1, f) != 1 || fread(&k, 4, 1, f) != 1 || fread(&L, 4, 1, f) != 1 || fread(&sc, 4, 1, f) != 1 || fread(&wg, 4, 1, f) != 1 || sz < 41 || sz > 256) {
            fclose(f); free(parent); free(desc); free(weight); free(leaf); return NULL; }
        uint8_t buf[256];
        for (unsigned i = 0; i < nb; i++) {
            if (fread(buf, sz, 1, f) != 1) break;
            GROW();
            memcpy(&parent[n], buf, 4); memcpy(desc + 32 * (size_t)n, buf + 4, 32);
            float wf; memcpy(&wf, buf + 36, 4); weight[n] = (double)wf; leaf[n] = buf[40] != 0; n++;
        }
    }
#undef GROW
    fclose(f);
    orc_voc *v = orc_voc_create(k, L, sc, wg, n, parent, desc, weight, leaf);
    free(parent); free(desc); free(weight); free(leaf);
    return v;
}
