// sgx_prof.h — optional per-kernel-class HIP-event timing shared by all modules (sgx_profile_* in sgx.h).
#pragma once
#include "sgx_rt.h"
enum { SGX_K_RESIZE = 0, SGX_K_FAST, SGX_K_OCTREE, SGX_K_ORIENT_DESC, SGX_K_STEREO, SGX_K_MOTION, SGX_K_MATCH, SGX_K_POSEOPT, SGX_K_UNPROJECT, SGX_K_MATCH_LOCAL, SGX_K_MAPGLUE, SGX_K_MASK, SGX_K_LK_PYR, SGX_K_LK_TRACK, SGX_K_FM_RANSAC, SGX_K_DET_FWD, SGX_K_DET_OUT, SGX_K_BA_LINEARIZE, SGX_K_BA_SCHUR, SGX_K_BA_SOLVE, SGX_K_BA_UPDATE, SGX_K_COUNT };
void sgx_prof_begin(int k, sgx_stream_t st);
void sgx_prof_end(int k, sgx_stream_t st);
