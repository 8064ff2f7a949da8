// Test probe (CPU tier): exposes the host-side weight split of sg_slam_amd/csrc/sgx_det_bf16.h (sgx_split_weights_bf16x3) through a C entry so that
// tests/test_bf16_split.py can check it against numpy.  Built with g++ -DSGX_EMU (the device part of the header is excluded).
#include "../../sg_slam_amd/csrc/sgx_block.h"
#include "../../sg_slam_amd/csrc/sgx_det_bf16.h"
extern "C" int probe_split_weights(const float *w, int outc, int K, int ldw, unsigned short *dst) { sgx_split_weights_bf16x3(w, outc, K, ldw, dst); return (K + 15) / 16; }
extern "C" unsigned short probe_bf16_rne(float x) { return sgx_bf16_rne(x); }
// sgx_rt.h declares these for the emulator build
thread_local sgx_dim3 blockIdx, blockDim, gridDim;
