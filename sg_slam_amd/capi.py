"""ctypes binding of the C-ABI declared in include/sgx.h (one Python method per entry point)."""
import ctypes as C
import numpy as np

KP_DTYPE = np.dtype([('x', 'f4'), ('y', 'f4'), ('size', 'f4'), ('angle', 'f4'), ('response', 'f4'),
                     ('octave', 'i4'), ('class_id', 'i4')])   # cv::KeyPoint / sgx_keypoint, 28 B


class SgxError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [('fx', C.c_float), ('fy', C.c_float), ('cx', C.c_float), ('cy', C.c_float), ('bf', C.c_float),
                ('min_x', C.c_float), ('max_x', C.c_float), ('min_y', C.c_float), ('max_y', C.c_float)]


class BaProblem(C.Structure):
    _fields_ = [('n_poses', C.c_int32), ('n_points', C.c_int32), ('n_edges', C.c_int32), ('poses', C.c_void_p), ('pose_fixed', C.c_void_p),
                ('points', C.c_void_p), ('edge_pose', C.c_void_p), ('edge_point', C.c_void_p), ('edge_obs', C.c_void_p), ('edge_info', C.c_void_p)]


class BaStats(C.Structure):
    _fields_ = [('iterations_first', C.c_int32), ('iterations_second', C.c_int32), ('free_poses', C.c_int32), ('reserved', C.c_int32),
                ('chi2_first', C.c_double), ('chi2_second', C.c_double)]


SGX_DET_MAX = 100


class Detection(C.Structure):
    _fields_ = [(k, C.c_float) for k in ('label', 'score', 'xmin', 'ymin', 'xmax', 'ymax')]


class Object2D(C.Structure):
    _fields_ = [('id', C.c_int32)] + [(k, C.c_float) for k in ('prob', 'x', 'y', 'w', 'h')]


class DetResult(C.Structure):
    _fields_ = [('n_raw', C.c_int32), ('raw', Detection * SGX_DET_MAX), ('n_objects', C.c_int32), ('objects', Object2D * SGX_DET_MAX),
                ('have_dynamic_for_mapping', C.c_int32), ('have_dynamic_for_rm_feature', C.c_int32),
                ('n_map_boxes', C.c_int32), ('map_boxes', Object2D * SGX_DET_MAX), ('n_rm_boxes', C.c_int32), ('rm_boxes', Object2D * SGX_DET_MAX)]


class FlowConfig(C.Structure):
    _fields_ = [('width', C.c_int32), ('height', C.c_int32), ('max_batch', C.c_int32), ('win_size', C.c_int32), ('max_level', C.c_int32),
                ('max_count', C.c_int32), ('epsilon', C.c_double)]


class TrackerConfig(C.Structure):
    _fields_ = [('streams', C.c_int32), ('width', C.c_int32), ('height', C.c_int32), ('nfeatures', C.c_int32), ('scale_factor', C.c_float), ('nlevels', C.c_int32),
                ('ini_th_fast', C.c_int32), ('min_th_fast', C.c_int32), ('cam', Camera), ('depth_map_factor', C.c_float), ('th_projection', C.c_float),
                ('local_map', C.c_int32), ('dynamic_mask', C.c_int32), ('max_boxes', C.c_int32), ('pipelined', C.c_int32)]


class OrbConfig(C.Structure):
    _fields_ = [('nfeatures', C.c_int32), ('scale_factor', C.c_float), ('nlevels', C.c_int32),
                ('ini_th_fast', C.c_int32), ('min_th_fast', C.c_int32), ('width', C.c_int32),
                ('height', C.c_int32), ('max_batch', C.c_int32)]


# every symbol include/sgx.h declares (tests/test_abi.py checks the export table against this list)
SYMBOLS = [
    'sgx_version', 'sgx_status_string', 'sgx_orb_create', 'sgx_orb_destroy', 'sgx_orb_keypoint_capacity', 'sgx_orb_get_tables',
    'sgx_orb_extract_batch_dev', 'sgx_orb_extract', 'sgx_orb_last_status', 'sgx_profile_enable', 'sgx_profile_num_classes', 'sgx_profile_class_name',
    'sgx_profile_read', 'sgx_match_project_frame_batch_dev', 'sgx_match_project_frame', 'sgx_match_project_local_batch_dev',
    'sgx_match_project_local', 'sgx_frame_stereo_from_rgbd_batch_dev', 'sgx_frame_unproject_batch_dev', 'sgx_frame_make_map_points_batch_dev',
    'sgx_frame_merge_matches_batch_dev', 'sgx_pose_optimization_batch_dev', 'sgx_pose_optimization', 'sgx_frame_motion_model_batch_dev',
    'sgx_local_bundle_adjustment', 'sgx_bundle_adjustment', 'sgx_det_create', 'sgx_det_destroy', 'sgx_det_info', 'sgx_det_detect',
    'sgx_det_detect_batch_dev', 'sgx_det_forward_batch_dev', 'sgx_frame_compact_keys_batch_dev', 'sgx_frame_gray_from_color_batch_dev',
    'sgx_det_gemm_mode', 'sgx_det_plan_step', 'sgx_dynamic_mask_batch_dev', 'sgx_tracker_create', 'sgx_tracker_destroy',
    'sgx_tracker_keypoint_capacity', 'sgx_tracker_record_bytes', 'sgx_tracker_set_initial_pose', 'sgx_tracker_step_dev', 'sgx_tracker_host_buffers',
    'sgx_tracker_step_host', 'sgx_tracker_wait_inputs', 'sgx_tracker_sync', 'sgx_tracker_read', 'sgx_tracker_snapshot_pose_dev',
    'sgx_tracker_snapshot_boxes_dev', 'sgx_tracker_pack_records_dev', 'sgx_tracker_frame_dev', 'sgx_tracker_extractor', 'sgx_flow_create',
    'sgx_dist_unique_id', 'sgx_dist_create', 'sgx_dist_destroy', 'sgx_dist_world', 'sgx_dist_gather_records',
    'sgx_flow_destroy', 'sgx_flow_reset', 'sgx_flow_levels', 'sgx_flow_lk_batch_dev', 'sgx_flow_lk', 'sgx_fundamental_ransac_batch_dev',
    'sgx_find_fundamental_mat', 'sgx_hamming_matrix', 'sgx_hamming_matrix_dev', 'sgx_match_search_for_triangulation', 'sgx_match_search_by_bow',
    'sgx_match_search_by_bow_kf', 'sgx_match_fuse_search', 'sgx_match_project_keyframe', 'sgx_match_fuse_search_sim3',
    'sgx_triangulate_new_map_points', 'sgx_mappoint_update_normal_and_depth', 'sgx_mappoint_distinctive_descriptors', 'sgx_sim3_solver_create',
    'sgx_sim3_solver_set_ransac_parameters', 'sgx_sim3_solver_iterate', 'sgx_sim3_solver_get_estimate', 'sgx_sim3_solver_destroy',
    'sgx_match_search_for_initialization', 'sgx_voc_load', 'sgx_voc_create', 'sgx_voc_info', 'sgx_voc_destroy', 'sgx_voc_transform',
    'sgx_voc_transform_batch_dev', 'sgx_voc_score', 'sgx_match_project_sim3', 'sgx_match_search_by_sim3', 'sgx_optimize_sim3',
    'sgx_optimize_essential_graph', 'sgx_correct_map_points',
]
# the test / tuning taps include/sgx_debug.h declares: exported by tests/taps/libsgx_taps.so and the emulator (-DSGX_DEBUG_TAPS) only, never by the product library
TAP_SYMBOLS = [
    'sgx_orb_debug_level_geometry', 'sgx_orb_debug_set_unfused_pyramid', 'sgx_orb_debug_read_level', 'sgx_orb_debug_read_candidates',
    'sgx_orb_debug_run_octree', 'sgx_pose_opt_debug_set_threads', 'sgx_ba_debug_set_solver', 'sgx_ba_debug_set_jobs', 'sgx_ba_debug_set_init', 'sgx_ba_debug_last_plan', 'sgx_det_debug_read_blob',
    'sgx_det_debug_detection_output', 'sgx_debug_flow_affine_batch_dev', 'sgx_det_debug_set_fusion', 'sgx_det_debug_set_legacy_kernels',
    'sgx_det_debug_set_block_fusion', 'sgx_det_debug_set_irb', 'sgx_det_debug_set_gemm', 'sgx_det_debug_time_ops', 'sgx_det_debug_run_step', 'sgx_flow_debug_read_level',
    'sgx_flow_debug_level_size', 'sgx_debug_corun_bf16',
]


def _vp(x):
    """device/host pointer from int, numpy array or anything with data_ptr()"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, 'data_ptr'):
        return C.c_void_p(x.data_ptr())
    return x.ctypes.data_as(C.c_void_p)


class SgxLib:
    def __init__(self, path):
        self.path = path
        self.dll = C.CDLL(path)
        d = self.dll
        d.sgx_version.restype = C.c_char_p
        d.sgx_status_string.restype = C.c_char_p
        d.sgx_status_string.argtypes = [C.c_int]
        d.sgx_orb_create.argtypes = [C.POINTER(OrbConfig), C.POINTER(C.c_void_p)]
        d.sgx_orb_destroy.argtypes = [C.c_void_p]
        d.sgx_orb_destroy.restype = None
        d.sgx_orb_keypoint_capacity.argtypes = [C.c_void_p]
        d.sgx_orb_get_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        d.sgx_orb_extract_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_void_p]
        d.sgx_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        d.sgx_orb_last_status.argtypes = [C.c_void_p, C.c_void_p]

        d.sgx_profile_enable.argtypes = [C.c_int]
        d.sgx_profile_class_name.restype = C.c_char_p
        d.sgx_profile_class_name.argtypes = [C.c_int]
        d.sgx_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]

        vp = C.c_void_p
        d.sgx_match_project_frame_batch_dev.argtypes = [C.c_int, C.c_int] + [vp] * 13 + [C.POINTER(Camera), vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp, vp]
        d.sgx_match_project_frame.argtypes = [C.c_int] + [vp] * 4 + [C.c_int] + [vp] * 7 + [C.POINTER(Camera), vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp]
        d.sgx_frame_stereo_from_rgbd_batch_dev.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp, vp]
        d.sgx_frame_unproject_batch_dev.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(Camera), vp, vp, vp]
        d.sgx_pose_optimization_batch_dev.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.POINTER(Camera), vp, vp, vp, vp]
        d.sgx_pose_optimization.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int, C.POINTER(Camera), vp, vp, vp]
        d.sgx_frame_motion_model_batch_dev.argtypes = [C.c_int, vp, vp, vp, vp, vp]
        d.sgx_local_bundle_adjustment.argtypes = [C.POINTER(BaProblem), C.POINTER(Camera), vp, vp, C.POINTER(BaStats)]
        d.sgx_bundle_adjustment.argtypes = [C.POINTER(BaProblem), C.POINTER(Camera), C.c_int, vp, C.c_int, C.POINTER(BaStats)]
        d.sgx_det_create.argtypes = [C.c_char_p, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(vp)]
        d.sgx_det_destroy.argtypes = [vp]; d.sgx_det_destroy.restype = None
        d.sgx_det_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        d.sgx_det_detect.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(DetResult)]
        d.sgx_det_detect_batch_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp]
        d.sgx_det_forward_batch_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), vp]
        d.sgx_det_gemm_mode.argtypes = [vp]
        d.sgx_tracker_create.argtypes = [C.POINTER(TrackerConfig), vp, C.POINTER(vp)]
        d.sgx_tracker_destroy.argtypes = [vp]; d.sgx_tracker_destroy.restype = None
        d.sgx_tracker_keypoint_capacity.argtypes = [vp]
        d.sgx_tracker_record_bytes.argtypes = [vp]
        d.sgx_tracker_set_initial_pose.argtypes = [vp, vp]
        d.sgx_tracker_step_dev.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, vp]
        d.sgx_tracker_host_buffers.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(vp)]
        d.sgx_tracker_step_host.argtypes = [vp, C.c_int, C.c_int]
        d.sgx_tracker_sync.argtypes = [vp]
        d.sgx_tracker_wait_inputs.argtypes = [vp, C.c_int]
        d.sgx_tracker_read.argtypes = [vp] * 10
        d.sgx_tracker_snapshot_pose_dev.argtypes = [vp, vp]
        d.sgx_tracker_snapshot_boxes_dev.argtypes = [vp, C.c_int, vp, vp]
        d.sgx_tracker_pack_records_dev.argtypes = [vp, vp, vp]
        d.sgx_dist_unique_id.argtypes = [vp]
        d.sgx_dist_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
        d.sgx_dist_destroy.argtypes = [vp]; d.sgx_dist_destroy.restype = None
        d.sgx_dist_world.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        d.sgx_dist_gather_records.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, vp]
        d.sgx_tracker_frame_dev.argtypes = [vp] + [C.POINTER(vp)] * 6
        d.sgx_tracker_extractor.argtypes = [vp]; d.sgx_tracker_extractor.restype = vp
        d.sgx_det_plan_step.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
        d.sgx_dynamic_mask_batch_dev.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp]
        d.sgx_frame_gray_from_color_batch_dev.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]
        d.sgx_frame_compact_keys_batch_dev.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]
        d.sgx_match_project_local.argtypes = [C.c_int] + [vp] * 5 + [C.c_int] + [vp] * 7 + [C.POINTER(Camera), vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, vp]
        d.sgx_match_project_local_batch_dev.argtypes = [C.c_int, C.c_int] + [vp] * 6 + [C.c_int] + [vp] * 8 + [C.POINTER(Camera), vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp]
        d.sgx_frame_make_map_points_batch_dev.argtypes = [C.c_int, C.c_int, C.c_int] + [vp] * 7 + [C.c_int] + [vp] * 7
        d.sgx_frame_merge_matches_batch_dev.argtypes = [C.c_int, C.c_int] + [vp] * 10
        d.sgx_flow_create.argtypes = [C.POINTER(FlowConfig), C.POINTER(vp)]
        d.sgx_flow_destroy.argtypes = [vp]; d.sgx_flow_destroy.restype = None
        d.sgx_flow_reset.argtypes = [vp]
        d.sgx_flow_levels.argtypes = [vp]
        d.sgx_flow_lk_batch_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, C.POINTER(C.c_int32), vp]
        d.sgx_flow_lk.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]
        d.sgx_fundamental_ransac_batch_dev.argtypes = [C.c_int, C.c_int] + [vp] * 6 + [C.c_int, C.c_double, C.c_double, vp, vp, vp, vp]
        d.sgx_find_fundamental_mat.argtypes = [vp, vp, C.c_int, C.c_double, C.c_double, vp, vp, vp]
        d.sgx_hamming_matrix.argtypes = [vp, C.c_int, vp, C.c_int, vp]
        d.sgx_optimize_essential_graph.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp]
        d.sgx_correct_map_points.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp, vp]
        d.sgx_optimize_sim3.argtypes = [C.c_int] + [vp] * 9 + [C.c_float, C.c_int, vp, vp, vp]
        d.sgx_match_project_keyframe.argtypes = [C.c_int] + [vp] * 4 + [C.c_int] + [vp] * 6 + [C.POINTER(Camera), vp, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, vp, vp]
        d.sgx_match_search_by_bow.argtypes = [C.c_int] + [vp] * 4 + [C.c_int] + [vp] * 3 + [C.c_float, C.c_int, vp, vp]
        d.sgx_match_fuse_search_sim3.argtypes = [C.c_int, vp, vp, vp, C.c_int] + [vp] * 6 + [vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp]
        d.sgx_match_project_sim3.argtypes = [C.c_int, vp, vp, vp, vp, C.c_int] + [vp] * 6 + [vp, vp, C.c_int, C.c_float, C.c_int, vp, vp]
        d.sgx_match_search_by_sim3.argtypes = ([C.c_int] + [vp] * 8) * 2 + [vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, C.c_float, vp, vp]
        d.sgx_match_search_for_initialization.argtypes = [C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]
        d.sgx_sim3_solver_create.argtypes = [C.c_int] + [vp] * 6 + [C.c_int, C.c_uint, vp]
        d.sgx_sim3_solver_set_ransac_parameters.argtypes = [vp, C.c_double, C.c_int, C.c_int]
        d.sgx_sim3_solver_iterate.argtypes = [vp, C.c_int] + [vp] * 7
        d.sgx_sim3_solver_get_estimate.argtypes = [vp] * 5
        d.sgx_sim3_solver_destroy.argtypes = [vp]; d.sgx_sim3_solver_destroy.restype = None
        d.sgx_mappoint_update_normal_and_depth.argtypes = [C.c_int] + [vp] * 6 + [C.c_int, vp, vp, vp]
        d.sgx_mappoint_distinctive_descriptors.argtypes = [C.c_int, vp, vp, vp, vp]
        d.sgx_triangulate_new_map_points.argtypes = [C.c_int, vp] + [C.c_int] + [vp] * 5 + [C.c_int] + [vp] * 5 + [vp, vp, vp, C.c_int, vp, vp, vp]
        d.sgx_voc_load.argtypes = [C.c_char_p, vp]
        d.sgx_voc_create.argtypes = [C.c_int] * 5 + [vp] * 5
        d.sgx_voc_info.argtypes = [vp] * 7
        d.sgx_voc_destroy.argtypes = [vp]; d.sgx_voc_destroy.restype = None
        d.sgx_voc_transform.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]
        d.sgx_voc_transform_batch_dev.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
        d.sgx_voc_score.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp]
        d.sgx_match_search_by_bow_kf.argtypes = [C.c_int] + [vp] * 4 + [C.c_int] + [vp] * 4 + [C.c_float, C.c_int, vp, vp]
        d.sgx_match_fuse_search.argtypes = [C.c_int] + [vp] * 4 + [C.c_int] + [vp] * 6 + [C.POINTER(Camera), vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp]
        d.sgx_hamming_matrix_dev.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
        d.sgx_match_search_for_triangulation.argtypes = [C.c_int] + [vp] * 6 + [C.c_int] + [vp] * 6 + [vp, C.POINTER(Camera), vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]

        # test / tuning taps (include/sgx_debug.h): present in tests/taps/libsgx_taps.so and the emulator only
        self.has_taps = hasattr(d, 'sgx_det_debug_read_blob')
        if self.has_taps:
            d.sgx_orb_debug_level_geometry.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int32)] * 3
            d.sgx_orb_debug_read_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            d.sgx_orb_debug_read_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
            d.sgx_orb_debug_run_octree.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
            d.sgx_orb_debug_set_unfused_pyramid.argtypes = [C.c_int]
            d.sgx_pose_opt_debug_set_threads.argtypes = [C.c_int]
            d.sgx_ba_debug_set_solver.argtypes = [C.c_int]
            d.sgx_ba_debug_set_jobs.argtypes = [C.c_int]
            d.sgx_ba_debug_set_init.argtypes = [C.c_int]
            d.sgx_ba_debug_last_plan.argtypes = [C.c_void_p]
            d.sgx_det_debug_detection_output.argtypes = [vp, vp, vp, C.c_int, C.POINTER(DetResult)]
            d.sgx_det_debug_read_blob.argtypes = [vp, C.c_char_p, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
            for nm in ('fusion', 'legacy_kernels', 'block_fusion', 'irb', 'gemm'):
                getattr(d, 'sgx_det_debug_set_' + nm).argtypes = [C.c_int]
            d.sgx_det_debug_time_ops.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
            d.sgx_det_debug_run_step.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
            d.sgx_debug_flow_affine_batch_dev.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp]
            d.sgx_flow_debug_read_level.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
            d.sgx_debug_corun_bf16.argtypes = [C.c_int, C.c_int, C.c_int, vp]
            d.sgx_flow_debug_level_size.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]

    def tap(self, name):
        """a test / tuning tap entry (include/sgx_debug.h); the product library has none"""
        if not self.has_taps:
            raise SgxError(f'{name}: {self.path} is the product build and has no test taps; use tests/taps/libsgx_taps.so (make -C sg_slam_amd/csrc taps)')
        return getattr(self.dll, name)

    def version(self):
        return self.dll.sgx_version().decode()

    def check(self, rc, what=''):
        if rc != 0:
            raise SgxError(f'{what}: {self.dll.sgx_status_string(rc).decode()} ({rc})')

    # per-kernel HIP-event timing (process-wide)
    def profile_enable(self, on=True):
        self.check(self.dll.sgx_profile_enable(int(on)))

    def profile_read(self, reset=True):
        n = self.dll.sgx_profile_num_classes()
        ms = np.zeros(n, 'f4'); cnt = np.zeros(n, 'i4')
        self.check(self.dll.sgx_profile_read(_vp(ms), _vp(cnt), int(reset)), 'sgx_profile_read')
        return {self.dll.sgx_profile_class_name(k).decode(): (float(ms[k]), int(cnt[k])) for k in range(n)}
