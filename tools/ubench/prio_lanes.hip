// round 6 micro-reproducer attempt: does a deterministic ALU-dense kernel return different results when kernels of a DIFFERENT stream priority run beside it?
// (the LK tracker did, for lanes 32-63 of a wave: profiles/r6_lk_priority_diagnosis.md).  usage: prio_lanes <mode> <reps> <prio: 0 equal, 1 mixed>
//   mode 1: v_dot2c chain + shift + v_mad_i32_i16 (the LK sample arithmetic)   2: DPP row reductions   3: ds_bpermute exchange   4: plain integer multiply-add (control)   5: all of them
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int mad_lo16(int d, uint32_t p, int acc) { asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,0,0,0]" : "+v"(acc) : "v"(d), "v"(p)); return acc; }
__device__ __forceinline__ int mad_hi16(int d, uint32_t p, int acc) { asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(acc) : "v"(d), "v"(p)); return acc; }
__device__ __forceinline__ int row_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, true); v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, true); v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, true);
    return v;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_victim(uint32_t *out, int iters, int mode, uint32_t seed)
{
    const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid, lane = tid & 63;
    uint32_t x = seed ^ (uint32_t)(gid * 2654435761u);
    int acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0;
    uint32_t px[8];
    for (int i = 0; i < 8; i++) { x = x * 1664525u + 1013904223u; px[i] = x; }
    for (int it = 0; it < iters; it++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t w0 = (x & 0x3fff) | ((x >> 2) & 0x3fff0000u), w1 = ((x >> 7) & 0x3fff) | ((x << 3) & 0x3fff0000u);
        if (mode == 1 || mode == 5) {
#pragma unroll
            for (int c = 0; c < 7; c++) {
                const s2 pa = __builtin_bit_cast(s2, __builtin_amdgcn_perm(px[c + 1], px[c], 0x0c010c00u)), pb = __builtin_bit_cast(s2, __builtin_amdgcn_perm(px[c], px[c + 1], 0x0c030c02u));
                const int d = __builtin_amdgcn_sdot2(pb, __builtin_bit_cast(s2, w1), __builtin_amdgcn_sdot2(pa, __builtin_bit_cast(s2, w0), 256 - (int)((px[c] >> 20) << 9), false), false) >> 9;
                acc1 = mad_lo16(d, px[(c + 3) & 7], acc1); acc2 = mad_hi16(d, px[(c + 3) & 7], acc2);
            }
        }
        if (mode == 2 || mode == 5) { acc3 += row_sum(acc1 & 0xFFFF) ^ row_sum((acc2 >> 16) + (int)(x & 255)); }
        if (mode == 3 || mode == 5) { acc4 += __shfl_xor(acc3 + (int)(x >> 24), 16, 64); }
        if (mode == 4) { acc1 = acc1 * 1103515245 + (int)w0; acc2 = acc2 * 69069 + (int)w1; }
        if (mode == 6 || mode == 7) {      // fp32 pairs: 6 = packed instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32), 7 = the same arithmetic with single instructions
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 u = { (float)(int)(x & 1023) * 0.125f, (float)(int)((x >> 10) & 1023) * 0.25f }, v = { __int_as_float(acc1) , __int_as_float(acc2) }, k = { 0.9990234375f, 1.0009765625f };
            if (!(v[0] == v[0]) || fabsf(v[0]) > 1e30f) v[0] = 1.f; if (!(v[1] == v[1]) || fabsf(v[1]) > 1e30f) v[1] = 2.f;
            if (mode == 6) {
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(k));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(u));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v) : "v"(u), "v"(k));
            } else {
                v[0] = v[0] * k[0]; v[1] = v[1] * k[1]; asm volatile("" : "+v"(v)); v[0] = v[0] + u[0]; v[1] = v[1] + u[1]; asm volatile("" : "+v"(v)); v[0] = __builtin_fmaf(u[0], k[0], v[0]); v[1] = __builtin_fmaf(u[1], k[1], v[1]);
            }
            acc1 = __float_as_int(v[0]); acc2 = __float_as_int(v[1]);
        }
        px[it & 7] ^= (uint32_t)(acc1 + acc3 + acc4) + x;
    }
    out[gid] = (uint32_t)acc1 ^ ((uint32_t)acc2 * 31u) ^ ((uint32_t)acc3 * 131u) ^ ((uint32_t)acc4 * 8191u) ^ px[lane & 7];
}
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// co-runner with the instruction classes of k_hrb: kind 1 = v_permlane32_swap, 2 = bf16 matrix products, 3 = both, 4 = v_pk_fma_f32 with scalar weights
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_busy2(float *o, int iters, int kind)
{
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
    f16v acc = {0}; bf8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((x >> i) & 7); b[i] = (__bf16)(float)((y >> i) & 3); }
    float f0 = 1.f, f1 = 2.f;
    for (int i = 0; i < iters; i++) {
        if (kind & 1) { const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); x = r[0] + i; y = r[1] ^ x; }
        if (kind & 2) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); }
        if (kind & 4) { f0 = f0 * 1.0001f + f1; f1 = f1 * 0.9999f + 1e-3f; }
    }
    float s = f0 + f1; for (int i = 0; i < 16; i++) s += acc[i];
    o[blockIdx.x * 256 + threadIdx.x] = s + (float)(x ^ y);
}
__global__ void __launch_bounds__(256) k_busy(float *o, int iters)
{
    __shared__ float sh[4096];
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = a + i;
    __syncthreads();
    for (int i = 0; i < iters; i++) { a = a * b + sh[(threadIdx.x * 17 + i) & 4095]; b = b * 0.99999f + 1e-6f; }
    o[blockIdx.x * 256 + threadIdx.x] = a + b;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 5, reps = argc > 2 ? atoi(argv[2]) : 200, mixed = argc > 3 ? atoi(argv[3]) : 1, kind = argc > 5 ? atoi(argv[5]) : 0;
    const int WG = argc > 4 ? atoi(argv[4]) : 1024, N = WG * 256;
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t sV[2], sB[2], sH[2];
    for (int i = 0; i < 2; i++) {
        CK(hipStreamCreateWithFlags(&sV[i], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB[i], hipStreamNonBlocking));
        if (mixed) CK(hipStreamCreateWithPriority(&sH[i], hipStreamNonBlocking, hi)); else CK(hipStreamCreateWithFlags(&sH[i], hipStreamNonBlocking));
    }
    uint32_t *d_out[2], *d_ref; float *d_busy; hipEvent_t ev[2];
    for (int i = 0; i < 2; i++) { CK(hipMalloc(&d_out[i], N * 4)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    CK(hipMalloc(&d_ref, N * 4)); CK(hipMalloc(&d_busy, 1024 * 256 * 4));
    std::vector<uint32_t> ref(N), got(N);
    hipLaunchKernelGGL(k_victim, dim3(WG), dim3(256), 0, sV[0], d_ref, 400, mode, 12345u); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), d_ref, N * 4, hipMemcpyDeviceToHost));
    long bad_lanes[64] = { 0 }; long bad = 0, bad_runs = 0;
    for (int r = 0; r < reps; r++) {
        for (int i = 0; i < 2; i++) {      // two "trackers": busy work on a side stream, the victim on the extraction stream, then work on the (high-priority) tracking stream behind an event
            if (kind) hipLaunchKernelGGL(k_busy2, dim3(512), dim3(256), 0, sB[i], d_busy, 3000, kind); else hipLaunchKernelGGL(k_busy, dim3(512), dim3(256), 0, sB[i], d_busy, 3000);
            hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, sV[i], d_busy + 65536, 500);
            hipLaunchKernelGGL(k_victim, dim3(WG), dim3(256), 0, sV[i], d_out[i], 400, mode, 12345u);
            CK(hipEventRecord(ev[i], sV[i])); CK(hipStreamWaitEvent(sH[i], ev[i], 0));
            for (int k = 0; k < 6; k++) hipLaunchKernelGGL(k_busy, dim3(2), dim3(256), 0, sH[i], d_busy + 131072 + 2048 * k, 2000);
        }
        CK(hipDeviceSynchronize());
        for (int i = 0; i < 2; i++) {
            CK(hipMemcpy(got.data(), d_out[i], N * 4, hipMemcpyDeviceToHost));
            long b = 0; for (int j = 0; j < N; j++) if (got[j] != ref[j]) { b++; bad_lanes[j & 63]++; }
            if (b) { bad += b; bad_runs++; }
        }
    }
    printf("co-runner kind %d, mode %d mixed %d reps %d: victim launches with a difference %ld of %d, differing lanes %ld; by lane quarter [0-15 16-31 32-47 48-63] = ", kind, mode, mixed, reps, bad_runs, 2 * reps, bad);
    for (int q = 0; q < 4; q++) { long s = 0; for (int l = 16 * q; l < 16 * q + 16; l++) s += bad_lanes[l]; printf("%ld ", s); }
    printf("\n");
    return 0;
}
