"""DBoW2 vocabulary transform (Frame::ComputeBoW) on the device against the oracle."""
import numpy as np
import pytest
import voc_cases as vc

pytestmark = pytest.mark.gpu


def test_voc_transform_gpu(gpulib, oracle):
    vc.check_transform(gpulib, oracle, n_cases=4)


def test_voc_files_and_score_gpu(gpulib, oracle, tmp_path):
    vc.check_files_and_score(gpulib, oracle, str(tmp_path))


def test_voc_bow_chain_gpu(gpulib, oracle):
    vc.check_bow_chain(gpulib, oracle)


def test_voc_transform_batch_dev_gpu(gpulib, oracle):
    """the device-resident form (B ragged frames per launch) gives the per-feature arrays of the one-frame entry"""
    import torch
    from sg_slam_amd.vocabulary import ORBVocabulary
    voc = vc.make_vocabulary(9, k=10, L=4)
    V = ORBVocabulary(gpulib).create(voc['k'], voc['L'], voc['parent'], voc['desc'], voc['weight'], voc['is_leaf'])
    B, cap = 5, 1024
    ns = np.array([1000, 0, 1, 777, 1024], 'i4')
    desc = np.zeros((B, cap, 32), np.uint8)
    for b in range(B): desc[b, :ns[b]] = vc.make_features(voc, 40 + b, int(ns[b])) if ns[b] else desc[b, :0]
    d_desc = torch.from_numpy(desc).cuda(); d_n = torch.from_numpy(ns).cuda()
    d_word = torch.full((B, cap), -7, dtype=torch.int32, device='cuda'); d_w = torch.zeros((B, cap), dtype=torch.float64, device='cuda'); d_fn = torch.full((B, cap), -7, dtype=torch.int32, device='cuda')
    V.transform_batch_dev(d_desc, cap * 32, d_n, B, cap, d_word, d_w, d_fn, levelsup=2)
    torch.cuda.synchronize()
    for b in range(B):
        _, _, fn, word = V.transform(desc[b, :ns[b]], 2)
        assert (d_fn[b, :ns[b]].cpu().numpy() == fn).all() and (d_word[b, :ns[b]].cpu().numpy() == word).all()
        assert (d_fn[b, ns[b]:].cpu().numpy() == -7).all()                   # rows beyond the count are left alone
    V.close()
