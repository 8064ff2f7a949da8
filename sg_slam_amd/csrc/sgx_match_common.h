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

