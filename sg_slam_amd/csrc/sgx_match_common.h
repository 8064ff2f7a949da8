// sgx_match_common.h — constants and device helpers shared by the matcher translation units (sgx_match.cpp, sgx_match2.cpp)
#pragma once
#include "sgx_rt.h"
#include "sgx_types.h"
#include "sgx_block.h"
#define SGX_TH_HIGH 100           /* ORBmatcher::TH_HIGH, ORBmatcher.cc:37 */
#define SGX_HISTO 30              /* ORBmatcher::HISTO_LENGTH, ORBmatcher.cc:39 */

// 256-bit Hamming distance == ORBmatcher::DescriptorDistance (ORBmatcher.cc:1649-1665; SWAR popcount == popcount)
SGX_DEV int sgx_hamming256(const uint32_t *a, const uint32_t *b)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const unsigned long long x = ((unsigned long long)(a[i] ^ b[i])) | ((unsigned long long)(a[i + 1] ^ b[i + 1]) << 32);
        d += (int)SGX_POPCLL(x);
    }
    return d;
}

// cv::gemm small-matrix path for 3x3 * 3x1 (+ c): float dot left-to-right, then (float)(t*alpha + beta*c) in double
SGX_DEV float sgx_gemm3(const float *arow, const float *b, float c)
{
    const float t = arow[0] * b[0] + arow[1] * b[1] + arow[2] * b[2];
    return (float)((double)t * 1.0 + 1.0 * (double)c);
}
