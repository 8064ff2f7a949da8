"""Kernel name -> bench.py per-kernel class (the sgx_profile_* classes of sgx_prof.h), shared by the PMC post-processing tools."""
CLASS = [('k_pyramid', 'pyramid_resize'), ('k_resize', 'pyramid_resize'), ('k_gray_from_color', 'pyramid_resize'), ('k_fast_cells', 'fast_cells'), ('k_octree', 'octree'),
         ('k_blur_levels', 'orient_desc'), ('k_orient_desc', 'orient_desc'),
         ('k_stereo_from_rgbd', 'stereo_from_rgbd'), ('k_motion_model', 'motion_model'), ('k_match_project_frame', 'match_project_frame'),
         ('k_match_project_local', 'match_project_local'), ('k_pose_opt', 'pose_opt'), ('k_unproject', 'unproject'),
         ('k_make_map_points', 'map_point_glue'), ('k_merge_matches', 'map_point_glue'), ('k_gather_xw', 'map_point_glue'),
         ('k_dynamic_mask', 'dynamic_mask'), ('k_compact_keys', 'dynamic_mask'),
         ('k_lk_copy', 'lk_pyramid'), ('k_lk_pyrdown', 'lk_pyramid'), ('k_lk_track', 'lk_track'), ('k_fm_ransac', 'fm_ransac'),
         ('k_det_preprocess', 'det_forward'), ('k_stem_pre', 'det_forward'), ('k_conv_pw', 'det_forward'), ('k_conv_kxk', 'det_forward'), ('k_conv_dw', 'det_forward'), ('k_conv_stem', 'det_forward'), ('k_binary', 'det_forward'), ('k_unary', 'det_forward'),
         ('k_copy_into', 'det_forward'), ('k_permute_hwc_into', 'det_forward'), ('k_softmax_rows', 'det_forward'), ('k_fused_block', 'det_forward'),
         ('k_det_class_nms', 'det_output'), ('k_det_merge', 'det_output')]


def classify(kernel_name):
    base = kernel_name.split('(')[0]
    if base.startswith('void '): base = base[5:]
    base = base.split('<')[0].strip()
    for pre, cls in CLASS:
        if base.startswith(pre):
            return cls
    return None


# dispatches per bench step of the kernels that run more than once per step (everything else: once); used to find out in how many of the profiled steps a class
# ran at all — the first step of a run has no previous frame, so the tracking-stage kernels (LK, RANSAC, matchers, pose optimisation ...) run in one step fewer
DISPATCHES_PER_STEP = {'k_pose_opt': 2, 'k_merge_matches': 2, 'k_lk_pyrdown': 3, 'k_resize': 7}


def steps_ran(dispatch_counts):
    """dispatch_counts: {kernel base name: dispatches} of ONE class -> number of steps in which the class ran"""
    best = 0.0
    for name, n in dispatch_counts.items():
        per = 1
        for pre, k in DISPATCHES_PER_STEP.items():
            if name.startswith(pre): per = k
        best = max(best, n / per)
    return max(1, int(round(best)))


def base_name(kernel_name):
    base = kernel_name.split('(')[0]
    if base.startswith('void '): base = base[5:]
    return base.split('<')[0].strip()
