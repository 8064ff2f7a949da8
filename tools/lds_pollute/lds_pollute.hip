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

// ---- canary (round 6): a workgroup fills `words` dwords of LDS with an address-derived pattern and keeps re-reading them for `spin` rounds; any word that changes is recorded
// (first 64 events: block, word index, expected, found, round).  Run beside a suspect kernel: nothing but the workgroup itself may write its LDS.
extern "C" __global__ void __launch_bounds__(256) k_canary(int words, int spin, uint32_t seed, uint32_t *events, int *nevents)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = seed ^ (uint32_t)(i * 2654435761u) ^ (blockIdx.x << 20);
    __syncthreads();
    for (int s = 0; s < spin; s++) {
        for (int i = threadIdx.x; i < words; i += 256) {
            const uint32_t want = seed ^ (uint32_t)(i * 2654435761u) ^ (blockIdx.x << 20), got = lds[i];
            if (got != want) {
                const int e = atomicAdd(nevents, 1);
                if (e < 64) { events[5 * e] = blockIdx.x; events[5 * e + 1] = i; events[5 * e + 2] = want; events[5 * e + 3] = got; events[5 * e + 4] = s; }
                lds[i] = want;
            }
        }
        __syncthreads();
    }
}
static uint32_t *g_ev = nullptr; static int *g_nev = nullptr; static hipStream_t g_cst = nullptr;
extern "C" int lds_canary_launch(int kbytes_x4, int blocks, int spin, uint32_t seed)      /* LDS size in units of 256 bytes; asynchronous on its own stream */
{
    if (!g_cst && hipStreamCreateWithFlags(&g_cst, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!g_ev) { if (hipMalloc(&g_ev, 64 * 5 * 4) != hipSuccess || hipMalloc(&g_nev, 4) != hipSuccess) return -2; (void)hipMemset(g_nev, 0, 4); (void)hipDeviceSynchronize(); }
    hipLaunchKernelGGL(k_canary, dim3(blocks), dim3(256), (size_t)kbytes_x4 * 256, g_cst, kbytes_x4 * 64, spin, seed, g_ev, g_nev);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" int lds_canary_read(uint32_t *events /* 64 x 5 */, int reset)
{
    int n = 0; if (!g_ev) return 0;
    (void)hipStreamSynchronize(g_cst); (void)hipMemcpy(&n, g_nev, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(events, g_ev, 64 * 5 * 4, hipMemcpyDeviceToHost);
    if (reset) (void)hipMemset(g_nev, 0, 4);
    return n;
}

// ---- canary 2 (round 6): the LDS access forms of the LK tracker, per wave and without workgroup barriers: a 36 x 32-byte tile per half-wave rewritten every round with a
// round-dependent byte pattern (dword stores by the half-wave's lanes), then read back by all its lanes as UNALIGNED 8-byte pairs (two rows) and checked byte for byte;
// plus a ds_bpermute exchange (lane ^ 16) of a known value.  events: [type 1 = tile read, 2 = bpermute][lane quarter] counters.
__device__ __forceinline__ uint32_t pat_byte(uint32_t seed, int tile, int round, int byte) { uint32_t h = seed ^ (uint32_t)(tile * 7919 + round * 104729 + byte) * 2654435761u; return (h >> 13) & 255u; }
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_canary2(int rounds, uint32_t seed, int *counts /* [2][4] */)
{
    __shared__ uint32_t tiles[8][293];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane >> 5, l = lane & 31;
    uint32_t *tile = tiles[wv * 2 + row]; const uint8_t *t8 = (const uint8_t *)tile; const int tidx = blockIdx.x * 8 + wv * 2 + row;
    int bad_tile = 0, bad_perm = 0;
    for (int r = 0; r < rounds; r++) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (int d = l; d < 288; d += 32) { const int b = 4 * d; tile[d] = pat_byte(seed, tidx, r, b) | (pat_byte(seed, tidx, r, b + 1) << 8) | (pat_byte(seed, tidx, r, b + 2) << 16) | (pat_byte(seed, tidx, r, b + 3) << 24); }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < 4; q++) {
            const int off = ((l * 37 + r * 11 + q * 263) % 1100);
            uint32_t a[2], b[2]; __builtin_memcpy(a, t8 + off, 8); __builtin_memcpy(b, t8 + off + 32, 8);
            for (int i = 0; i < 8; i++) {
                const uint32_t ga = (a[i >> 2] >> (8 * (i & 3))) & 255u, gb = (b[i >> 2] >> (8 * (i & 3))) & 255u;
                bad_tile += (ga != pat_byte(seed, tidx, r, off + i)) + (gb != pat_byte(seed, tidx, r, off + 32 + i));
            }
        }
        const int v = (int)((uint32_t)(lane + 64 * r) * 2246822519u), got = __shfl_xor(v, 16, 64), want = (int)((uint32_t)((lane ^ 16) + 64 * r) * 2246822519u);
        bad_perm += got != want;
    }
    if (bad_tile) atomicAdd(&counts[lane >> 4], bad_tile);
    if (bad_perm) atomicAdd(&counts[4 + (lane >> 4)], bad_perm);
}
static int *g_c2 = nullptr;
extern "C" int lds_canary2_launch(int blocks, int rounds, uint32_t seed)
{
    if (!g_cst && hipStreamCreateWithFlags(&g_cst, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!g_c2) { if (hipMalloc(&g_c2, 32) != hipSuccess) return -2; (void)hipMemset(g_c2, 0, 32); (void)hipDeviceSynchronize(); }
    hipLaunchKernelGGL(k_canary2, dim3(blocks), dim3(256), 0, g_cst, rounds, seed, g_c2);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
extern "C" int lds_canary2_read(int *counts8, int reset)
{
    if (!g_c2) return -1;
    (void)hipStreamSynchronize(g_cst); (void)hipMemcpy(counts8, g_c2, 32, hipMemcpyDeviceToHost);
    if (reset) (void)hipMemset(g_c2, 0, 32);
    return 0;
}

// ---- co-runner with selectable instruction classes of k_hrb (round 6): kind bits 1 v_permlane32_swap, 2 bf16 matrix products (32x32x16), 4 plain fp32 FMAs, 8 LDS stores (float2, the
// whole dynamic allocation), 16 v_pk_fma_f32 with a scalar-register operand, 32 fp32 matrix products (32x32x2).  Asynchronous on its own stream.
typedef float cr_f16v __attribute__((ext_vector_type(16)));
typedef __bf16 cr_bf8 __attribute__((ext_vector_type(8)));
typedef float cr_f2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_corun(float *o, int iters, int kind, int lds_words, const float *wts)
{
    extern __shared__ float cl[];
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
    cr_f16v acc = {0}, acc2 = {0}; cr_bf8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((x >> i) & 7); b[i] = (__bf16)(float)((y >> i) & 3); }
    float f0 = 1.f, f1 = 2.f; cr_f2 pk = {1.f, 2.f}, pv = {0.5f, 0.25f};
    const cr_f2 ws = *(const cr_f2 *)(wts + 2 * (blockIdx.x & 7));      // wave-uniform: a scalar load
    for (int i = 0; i < iters; i++) {
        if (kind & 1) { const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); x = r[0] + i; y = r[1] ^ x; }
        if (kind & 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if (kind & 4) { f0 = f0 * 1.0001f + f1; f1 = f1 * 0.9999f + 1e-3f; }
        if (kind & 8) { const int w = (threadIdx.x * 2 + i * 514) % (lds_words - 2); *(cr_f2 *)(cl + (w & ~1)) = cr_f2{f0 + i, f1}; }
        if (kind & 16) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(pk) : "s"(ws), "v"(pv));
        if (kind & 32) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, f1, acc2, 0, 0, 0);
    }
    float s = f0 + f1 + pk[0] + pk[1]; for (int i = 0; i < 16; i++) s += acc[i] + acc2[i];
    if (kind & 8) { __syncthreads(); s += cl[threadIdx.x]; }
    o[blockIdx.x * 256 + threadIdx.x] = s + (float)(x ^ y);
}
static float *g_cro = nullptr, *g_crw = nullptr; static hipStream_t g_crs = nullptr;
extern "C" int corun_launch(int blocks, int iters, int kind, int lds_kb, int launches)
{
    if (!g_crs && hipStreamCreateWithFlags(&g_crs, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!g_cro) { if (hipMalloc(&g_cro, 4096 * 256 * 4) != hipSuccess || hipMalloc(&g_crw, 64) != hipSuccess) return -2; (void)hipMemset(g_crw, 0, 64); (void)hipDeviceSynchronize(); }
    if (lds_kb > 64 && hipFuncSetAttribute((const void *)k_corun, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024) != hipSuccess) return -4;
    if (blocks > 4096) blocks = 4096;
    for (int l = 0; l < launches; l++) hipLaunchKernelGGL(k_corun, dim3(blocks), dim3(256), (size_t)lds_kb * 1024, g_crs, g_cro, iters, kind, lds_kb * 256, g_crw);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---- CU-masked streams (round 6): is the LK difference an interference inside a CU or a chip-level effect?  half 0 / 1 = the first / second 128 bits of the CU mask (disjoint CU sets
// whatever the bit -> CU mapping is), -1 = no mask.  corun_set_stream makes corun_launch use the given stream.
extern "C" void *corun_make_stream(int half)
{
    hipStream_t s = nullptr;
    if (half < 0) return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? (void *)s : nullptr;
    uint32_t mask[8]; for (int i = 0; i < 8; i++) mask[i] = (i / 4 == half) ? 0xFFFFFFFFu : 0u;
    return hipExtStreamCreateWithCUMask(&s, 8, mask) == hipSuccess ? (void *)s : nullptr;
}
extern "C" void corun_set_stream(void *s) { g_crs = (hipStream_t)s; }
extern "C" int corun_sync() { return g_crs ? (int)hipStreamSynchronize(g_crs) : 0; }
