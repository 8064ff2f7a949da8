"""Per-frame glue between the accelerated stages, mirroring the reference's Frame helpers
(src/sg-slam/src/Frame.cc:893-932) over the C-ABI batch entry points.  Pointers may be torch
CUDA tensors (product) or numpy arrays (kernel-logic emulator in tests)."""
import ctypes as C
from .capi import _vp
from .matcher import camera_struct


def stereo_from_rgbd_batch(lib, batch, cap, d_keys, d_n, d_depth_u16, width, height, depth_map_factor, bf, d_uright, d_zdepth, stream=None):
    lib.check(lib.dll.sgx_frame_stereo_from_rgbd_batch_dev(batch, cap, _vp(d_keys), _vp(d_n), _vp(d_depth_u16), width, height,
                                                           float(depth_map_factor), float(bf), _vp(d_uright), _vp(d_zdepth), _vp(stream)),
              'sgx_frame_stereo_from_rgbd_batch_dev')


def unproject_batch(lib, batch, cap, d_keys, d_n, d_zdepth, d_Tcw, cam, d_xw, d_has, stream=None):
    cs = camera_struct(cam)
    lib.check(lib.dll.sgx_frame_unproject_batch_dev(batch, cap, _vp(d_keys), _vp(d_n), _vp(d_zdepth), _vp(d_Tcw), C.byref(cs), _vp(d_xw),
                                                    _vp(d_has), _vp(stream)), 'sgx_frame_unproject_batch_dev')


def make_map_points_batch(lib, batch, cap, half, d_keys, d_n, d_xw, d_has, d_desc, d_Tcw, scale_factors, d_m_xw, d_m_normal, d_m_min, d_m_max,
                          d_m_desc, d_m_skip, stream=None):
    """MapPoint::MapPoint(Pos, pMap, pFrame, idxF) (MapPoint.cc:45-67) for every keypoint with depth -> slice `half` of a 2*cap ring."""
    import numpy as np
    sf = np.ascontiguousarray(scale_factors, 'f4')
    lib.check(lib.dll.sgx_frame_make_map_points_batch_dev(batch, cap, int(half), _vp(d_keys), _vp(d_n), _vp(d_xw), _vp(d_has), _vp(d_desc), _vp(d_Tcw),
                                                          _vp(sf), len(sf), _vp(d_m_xw), _vp(d_m_normal), _vp(d_m_min), _vp(d_m_max), _vp(d_m_desc),
                                                          _vp(d_m_skip), _vp(stream)), 'sgx_frame_make_map_points_batch_dev')


def merge_matches_batch(lib, batch, cap, d_n, d_match_last, d_outlier_last, d_match_local=None, d_xw_last=None, d_m_xw=None, d_merged=None,
                        d_cur_mp_obs=None, d_xw_all=None, stream=None):
    """mvpMapPoints after TrackWithMotionModel (+ SearchLocalPoints) as one index into [last.xw | local-map ring] (Tracking.cc:941-956)."""
    lib.check(lib.dll.sgx_frame_merge_matches_batch_dev(batch, cap, _vp(d_n), _vp(d_match_last), _vp(d_outlier_last), _vp(d_match_local), _vp(d_xw_last),
                                                        _vp(d_m_xw), _vp(d_merged), _vp(d_cur_mp_obs), _vp(d_xw_all), _vp(stream)),
              'sgx_frame_merge_matches_batch_dev')


def compact_keys_batch(lib, batch, cap, d_keys, d_desc, d_n, d_keep, d_have_dynamic, nfeatures, d_keys_out, d_desc_out, d_n_out, stream=None):
    """The erase step of Frame::RmDynamicPointWithSemanticAndGeometry (Frame.cc:556-604): order-preserving removal of masked keypoints and
    their descriptor rows, with the "restore all when < 0.1*nFeatures survive and a dynamic object is present" rule."""
    lib.check(lib.dll.sgx_frame_compact_keys_batch_dev(batch, cap, _vp(d_keys), _vp(d_desc), _vp(d_n), _vp(d_keep), _vp(d_have_dynamic), int(nfeatures),
                                                       _vp(d_keys_out), _vp(d_desc_out), _vp(d_n_out), _vp(stream)), 'sgx_frame_compact_keys_batch_dev')


def gray_from_color_batch(lib, batch, width, height, d_src, src_pitch, channels, blue_first, d_gray, gray_pitch, stream=None):
    """cvtColor(..., CV_{RGB,BGR,RGBA,BGRA}2GRAY) of Tracking::GrabImageRGBD (Tracking.cc:214-227) on device images."""
    lib.check(lib.dll.sgx_frame_gray_from_color_batch_dev(batch, width, height, _vp(d_src), src_pitch, channels, 1 if blue_first else 0, _vp(d_gray), gray_pitch, _vp(stream)),
              'sgx_frame_gray_from_color_batch_dev')
