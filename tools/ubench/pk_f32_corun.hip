// micro-reproducer (round 6): packed fp32 instructions of one wave beside a dense stream of bf16 matrix products of ANOTHER wave on the same CU.
// Victim: every lane of every wave runs the same chain on the same (lane-uniform) inputs — the float step arithmetic of the LK tracker (b = (hi * 65536 + lo) * 2^-20 for two sums,
// the 2 x 2 solve, position update) — so within a wave all 64 lanes must end with identical bits; the kernel counts lanes that differ from lane 0, by 16-lane quarter.
// Built twice from this one source: default flags (the SLP vectoriser pairs the two components into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and -fno-slp-vectorize (single
// v_fma_f32 / v_mul_f32 / v_add_f32).  Co-runner: v_mfma_f32_32x32x16_bf16 back to back (corun = 1) or plain fp32 FMAs (corun = 2) or nothing (0), on a second stream.
// usage: pk_f32_corun [corun] [reps] [victim blocks] [iterations]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_corun(float *o, int iters, int kind)
{
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
    f16v acc = {0}; bf8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)((x >> i) & 7); b[i] = (__bf16)(float)((y >> i) & 3); }
    float f0 = 1.f, f1 = 2.f;
    for (int i = 0; i < iters; i++) {
        if (kind == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        else { f0 = f0 * 1.0001f + f1; f1 = f1 * 0.9999f + 1e-3f; }
    }
    float s = f0 + f1; for (int i = 0; i < 16; i++) s += acc[i];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_victim(const int *in, int iters, float A11, float A12, float A22, float D, unsigned *bad /* [4] lanes by quarter, [4] = waves */, float *sink)
{
    const int lane = threadIdx.x & 63;
    float nx = 100.25f, ny = 50.75f; int carry = lane * 0;      /* `carry` keeps the integers in vector registers */
    for (int i = 0; i < iters; i++) {
        const int *p = in + 4 * (i & 255);
        const int lo1 = p[0] + carry, hi1 = p[1] + carry, lo2 = p[2] + carry, hi2 = p[3] + carry;
        const float b1 = fmaf((float)hi1, 65536.0f, (float)lo1) * (1.f / (1 << 20)), b2 = fmaf((float)hi2, 65536.0f, (float)lo2) * (1.f / (1 << 20));
        const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        const float f2 = dx * dx + dy * dy;
        if (f2 > 1e30f) carry = 1;      /* never: keeps f2 alive as in the tracker's convergence test */
    }
    const uint32_t ux = __float_as_uint(nx), uy = __float_as_uint(ny);
    const bool differs = ux != (uint32_t)__builtin_amdgcn_readfirstlane((int)ux) || uy != (uint32_t)__builtin_amdgcn_readfirstlane((int)uy);
    if (differs) atomicAdd(&bad[lane >> 4], 1u);
    if (__any(differs) && lane == 0) atomicAdd(&bad[4], 1u);
    if (nx == 12345.678f) sink[0] = ny;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int corun = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 50, vb = argc > 3 ? atoi(argv[3]) : 2048, iters = argc > 4 ? atoi(argv[4]) : 4000;
    hipStream_t sV, sC; CK(hipStreamCreateWithFlags(&sV, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));
    int h_in[1024]; srand(7);
    for (int i = 0; i < 256; i++) { h_in[4 * i] = rand() & 0x3fffff; h_in[4 * i + 1] = (rand() & 0x3ff) - 512; h_in[4 * i + 2] = rand() & 0x3fffff; h_in[4 * i + 3] = (rand() & 0x3ff) - 512; }
    int *d_in; unsigned *d_bad; float *d_o, *d_sink;
    CK(hipMalloc(&d_in, sizeof h_in)); CK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice)); CK(hipMalloc(&d_bad, 32)); CK(hipMemset(d_bad, 0, 32)); CK(hipMalloc(&d_o, 4096 * 256 * 4)); CK(hipMalloc(&d_sink, 4));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; r++) {
        if (corun) for (int l = 0; l < 3; l++) hipLaunchKernelGGL(k_corun, dim3(1024), dim3(256), 0, sC, d_o, 2000, corun);
        hipLaunchKernelGGL(k_victim, dim3(vb), dim3(256), 0, sV, d_in, iters, 812.5f, -37.25f, 640.75f, 1.f / (812.5f * 640.75f - 37.25f * 37.25f), d_bad, d_sink);
        CK(hipDeviceSynchronize());
    }
    unsigned h_bad[8]; CK(hipMemcpy(h_bad, d_bad, 32, hipMemcpyDeviceToHost));
    printf("co-runner %s, %d victim launches x %d workgroups x %d chain steps: waves with a lane that differs from lane 0: %u of %ld; differing lanes by quarter [0-15 16-31 32-47 48-63] = %u %u %u %u\n",
           corun == 1 ? "bf16 matrix products" : corun == 2 ? "fp32 FMAs" : "none", reps, vb, iters, h_bad[4], (long)reps * vb * 4, h_bad[0], h_bad[1], h_bad[2], h_bad[3]);
    return 0;
}
