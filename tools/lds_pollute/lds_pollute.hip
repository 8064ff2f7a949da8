// test tool (round 6): fills the LDS of every CU with a pattern, over and over, on a side stream.  LDS is not cleared between kernels, so a kernel that reads LDS it has not
// written sees whatever the previous workgroup on that CU left: with this running beside a test, such a read turns into a visible difference.  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void __launch_bounds__(256) k_pollute(uint32_t pattern, int words, int spin)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern ^ (uint32_t)(i * 2654435761u) ^ (blockIdx.x << 7);
    __syncthreads();
    uint32_t acc = 0;
    for (int s = 0; s < spin; s++) acc += lds[(threadIdx.x * 33 + s) % words];
    if (acc == 0x12345678u && pattern == 1) lds[0] = acc;      // keep the reads
}
static hipStream_t g_st = nullptr;
extern "C" int lds_pollute(uint32_t pattern, int kbytes, int launches, int blocks)
{
    if (!g_st && hipStreamCreateWithFlags(&g_st, hipStreamNonBlocking) != hipSuccess) return -1;
    if (kbytes > 64 && hipFuncSetAttribute((const void *)k_pollute, hipFuncAttributeMaxDynamicSharedMemorySize, kbytes * 1024) != hipSuccess) return -2;
    for (int i = 0; i < launches; i++) hipLaunchKernelGGL(k_pollute, dim3(blocks), dim3(256), (size_t)kbytes * 1024, g_st, pattern + i, kbytes * 256, 16);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" int lds_pollute_sync() { return g_st ? (int)hipStreamSynchronize(g_st) : 0; }
