"""ORBVocabulary — Python mirror of DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (src/sg-slam/include/ORBVocabulary.h:31-32) over the C ABI.

The reference loads the vocabulary once (System.cc:65-80) and calls transform() from Frame::ComputeBoW / KeyFrame::ComputeBoW (Frame.cc:422-429, KeyFrame.cc:60-69)
and score() from KeyFrameDatabase / LoopClosing.  The per-feature tree descent runs on the device (sgx_voc_transform)."""
import ctypes as C
import numpy as np
from . import load
from .capi import _vp


class ORBVocabulary:
    def __init__(self, lib=None):
        self.lib = lib or load()
        self.h = C.c_void_p()

    def _set(self, h):
        self.close(); self.h = h
        info = [C.c_int32() for _ in range(6)]
        self.lib.check(self.lib.dll.sgx_voc_info(self.h, *[C.byref(x) for x in info]), 'sgx_voc_info')
        self.k, self.L, self.scoring, self.weighting, self.nnodes, self.nwords = (int(x.value) for x in info)

    def loadFromTextFile(self, filename):
        assert filename.endswith('.txt'), 'the text loader is selected by the .txt suffix (System.cc:69-73)'
        return self._load(filename)

    def loadFromBinaryFile(self, filename):
        assert not filename.endswith('.txt')
        return self._load(filename)

    def _load(self, filename):
        h = C.c_void_p()
        rc = self.lib.dll.sgx_voc_load(filename.encode(), C.byref(h))
        if rc != 0: return False                                   # the reference's loaders return false on a bad file
        self._set(h); return True

    def create(self, k, L, parent, desc, weight, is_leaf, scoring=0, weighting=0):
        """the tree from flat arrays (node 0 = root): what both file formats hold"""
        parent = np.ascontiguousarray(parent, 'i4'); desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, 'f8'); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        h = C.c_void_p()
        self.lib.check(self.lib.dll.sgx_voc_create(k, L, scoring, weighting, len(parent), _vp(parent), _vp(desc), _vp(weight), _vp(is_leaf), C.byref(h)), 'sgx_voc_create')
        self._set(h); return self

    def size(self):
        return self.nwords

    def empty(self):
        return self.nwords == 0

    def transform(self, descriptors, levelsup=4):
        """transform(features, BowVector, FeatureVector, levelsup): (bow_ids, bow_weights, feat_node[n], feat_word[n]).  feat_node is mFeatVec flattened: the node id of
        every feature (-1 = stopped word), the form the BoW matchers take."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32); n = len(d)
        ids = np.zeros(max(n, 1), 'i4'); w = np.zeros(max(n, 1), 'f8'); nb = np.zeros(1, 'i4'); fn = np.full(max(n, 1), -1, 'i4'); fw = np.full(max(n, 1), -1, 'i4')
        self.lib.check(self.lib.dll.sgx_voc_transform(self.h, n, _vp(d), int(levelsup), _vp(ids), _vp(w), _vp(nb), _vp(fn), _vp(fw)), 'sgx_voc_transform')
        return ids[:nb[0]].copy(), w[:nb[0]].copy(), fn[:n].copy(), fw[:n].copy()

    def score(self, v1, v2):
        """score(BowVector, BowVector) (L1Scoring): v = (ids ascending, weights)"""
        i1 = np.ascontiguousarray(v1[0], 'i4'); w1 = np.ascontiguousarray(v1[1], 'f8'); i2 = np.ascontiguousarray(v2[0], 'i4'); w2 = np.ascontiguousarray(v2[1], 'f8')
        s = C.c_double()
        self.lib.check(self.lib.dll.sgx_voc_score(self.h, len(i1), _vp(i1), _vp(w1), len(i2), _vp(i2), _vp(w2), C.byref(s)), 'sgx_voc_score')
        return float(s.value)

    def transform_batch_dev(self, d_desc, desc_pitch, d_n, batch, cap, d_word, d_weight, d_feat_node, levelsup=4, stream=None):
        self.lib.check(self.lib.dll.sgx_voc_transform_batch_dev(self.h, _vp(d_desc), desc_pitch, _vp(d_n), batch, cap, int(levelsup), _vp(d_word), _vp(d_weight), _vp(d_feat_node),
                                                                 _vp(stream) if stream is not None else None), 'sgx_voc_transform_batch_dev')

    def close(self):
        if self.h: self.lib.dll.sgx_voc_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try: self.close()
        except Exception: pass
