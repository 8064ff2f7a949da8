// sgx_voc_kernels.h — the DBoW2 vocabulary descent of Frame::ComputeBoW / KeyFrame::ComputeBoW (src/sg-slam/src/Frame.cc:422-429:
// mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)), per feature:
//   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)    src/sg-slam/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1231-1273
//   FORB::distance                                                              src/sg-slam/Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101
// One thread per feature walks the k-ary tree from the root: at every level the Hamming distance to each child's descriptor, first minimum wins (strict '<' in child order).
// The tree lives in HBM as flat arrays (children as CSR in insertion order, descriptors as 8 dwords per node); ORBvoc (k = 10, L = 6, 1.1 M nodes) is 35 MB of descriptors,
// a frame's 1 000 features touch 60 k of them.  B frames per launch (grid.y), ragged counts.
#pragma once
#include "sgx_match_common.h"

struct SgxVocDev {
    int L, nnodes;
    const int *child_start, *child_idx, *word_id;
    const uint32_t *desc;
    const double *weight;
};

SGX_KERNEL(256) k_voc_transform(SgxVocDev V, int levelsup, const uint8_t *desc, size_t desc_pitch, const int *n_arr, int n_fixed, int cap,
                                int *word_id, double *weight, int *feat_node)
{
    SGX_THREADS_BEGIN(tid)
    const int b = (int)blockIdx.y, i = (int)blockIdx.x * 256 + tid;
    const int n = n_arr ? min(n_arr[b], cap) : n_fixed;
    if (i < n) {
        const uint32_t *f = (const uint32_t *)(desc + (size_t)b * desc_pitch) + (size_t)i * 8;
        uint32_t fv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) fv[k] = f[k];
        const int nid_level = V.L - levelsup;
        int nid = 0; bool nid_set = nid_level <= 0;
        int final_id = 0, level = 0;
        for (;;) {
            ++level;
            const int s = V.child_start[final_id], e = V.child_start[final_id + 1];
            int best = V.child_idx[s], best_d = sgx_hamming256(fv, V.desc + (size_t)best * 8);
            for (int c = s + 1; c < e; c++) {
                const int id = V.child_idx[c];
                const int d = sgx_hamming256(fv, V.desc + (size_t)id * 8);
                if (d < best_d) { best_d = d; best = id; }
            }
            final_id = best;
            if (level == nid_level) { nid = final_id; nid_set = true; }
            if (V.child_start[final_id + 1] <= V.child_start[final_id]) break;       // isLeaf()
        }
        if (!nid_set) nid = final_id;             // the reference leaves *nid unassigned when the descent ends above level L - levelsup; the leaf is the evident intent
        const double w = V.weight[final_id];
        const size_t o = (size_t)b * cap + i;
        word_id[o] = V.word_id[final_id]; weight[o] = w; feat_node[o] = w > 0 ? nid : -1;
    }
    SGX_THREADS_END
}
