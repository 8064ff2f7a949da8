// micro-reproducer (round 6): packed fp32 instructions of one wave beside a dense stream of bf16 matrix products of ANOTHER wave on the same CU.
// Victim: every lane of every wave runs the same chain on the same (lane-uniform) inputs — the float step arithmetic of the LK tracker (b = (hi * 65536 + lo) * 2^-20 for two sums,
// the 2 x 2 solve, position update) — so within a wave all 64 lanes must end with identical bits; the kernel counts lanes that differ from lane 0, by 16-lane quarter.
// Built twice from this one source: default flags (the SLP vectoriser pairs the two components into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and -fno-slp-vectorize (single
// v_fma_f32 / v_mul_f32 / v_add_f32).  Co-runner: v_mfma_f32_32x32x16_bf16 back to back (corun = 1) or plain fp32 FMAs (corun = 2) or nothing (0), on a second stream.
// usage: pk_f32_corun [corun] [reps] [victim blocks] [iterations] [rows: bit mask of the 16-lane rows that run the chain, default 15] [mode: 0 = the tracker's arithmetic, 1-6 = single forms below] [s_nop operand between links]
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
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_victim(const int *in, int iters, float A11, float A12, float A22, float D, unsigned *bad /* [4] lanes by quarter, [4] = waves */, float *sink, int rows /* bit q: the 16-lane row q of every wave runs the chain (the others sit the loop out: partial EXEC, as a finished keypoint's half-wave in the tracker) */)
{
    const int lane = threadIdx.x & 63;
    float nx = 100.25f, ny = 50.75f; int carry = lane * 0;      /* `carry` keeps the integers in vector registers */
    const bool on = (rows >> (lane >> 4)) & 1;
    if (on)
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
    const int first = __builtin_ctz(rows & 15) * 16;      /* the first lane that ran the chain */
    const bool differs = on && (ux != (uint32_t)__builtin_amdgcn_readlane((int)ux, first) || uy != (uint32_t)__builtin_amdgcn_readlane((int)uy, first));
    if (differs) atomicAdd(&bad[lane >> 4], 1u);
    if (__any(differs) && lane == 0) atomicAdd(&bad[4], 1u);
    if (nx == 12345.678f) sink[0] = ny;
}
// ---- single instruction forms (mode > 0): a dependent chain of ONE packed-fp32 form in inline assembly, every lane on the same values; `W` = the s_nop operand between two links
// (the compiler puts s_nop 0 = one wait state between dependent packed ops).  mode 1: v_pk_mul_f32 v, v, v   2: v_pk_fma_f32 v, v, s[..] op_sel_hi:[1,0,1], v   3: v_pk_add_f32 v, v, v
// 4: v_pk_mul_f32 links separated by eight independent single-rate instructions instead of s_nop   5: v_mul_f32 pairs (not packed)   6: v_pk_mul_f32 with op_sel:[0,1] op_sel_hi:[1,0] (crossed halves)   7 / 8: the detector's tap forms (v_pk_fma_f32 s, v, v with the vector operand's high / low half broadcast)   9: broadcasts in v_pk_mul_f32   10: v_pk_mul_f32 with neg_lo / neg_hi only
typedef float vf2 __attribute__((ext_vector_type(2)));
template <int MODE, int W>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_form(int iters, float k1, float k2, unsigned *bad, float *sink)
{
    const int lane = threadIdx.x & 63;
    vf2 x = { 1.25f + 0.f * lane, 2.5f + 0.f * lane }, ka = { k1, k2 }, kb = { k2, k1 }, c = { 1e-3f, -1e-3f };
    vf2 ks; ks[0] = k1; ks[1] = k2; vf2 cn = { 2e-3f, -1e-3f };      /* c[1] = -1e-3: mode 7 adds and removes the same products */
    if (MODE == 8) { cn[0] = -1e-3f; cn[1] = 5.f; }
    float pad = 1.f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 1) { asm volatile("v_pk_mul_f32 %0, %0, %1\n s_nop %3\n v_pk_mul_f32 %0, %0, %2\n s_nop %3" : "+v"(x) : "v"(ka), "v"(kb), "n"(W)); }
        if (MODE == 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]\n s_nop %3\n v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n s_nop %3" : "+v"(x) : "s"(ks), "v"(c), "n"(W)); x[0] *= k2; x[1] *= k2; }
        if (MODE == 3) { asm volatile("v_pk_add_f32 %0, %0, %1\n s_nop %3\n v_pk_add_f32 %0, %0, %2\n s_nop %3" : "+v"(x) : "v"(ka), "v"(c), "n"(W)); x[0] *= 0.5f; x[1] *= 0.5f; }
        if (MODE == 4) { asm volatile("v_pk_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n"
                                      "v_pk_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1\n v_mul_f32 %1, %1, %1" : "+v"(x), "+v"(pad) : "v"(ka), "v"(kb)); }
        if (MODE == 5) { asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %3\n s_nop %4\n v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %2\n s_nop %4" : "+v"(x[0]), "+v"(x[1]) : "v"(k1), "v"(k2), "n"(W)); }
        if (MODE == 7) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n s_nop %4\n v_pk_fma_f32 %0, %1, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n s_nop %4" : "+v"(x) : "s"(ks), "v"(c), "v"(cn), "n"(W)); }      /* the detector's depthwise tap: scalar weight pair x one pixel (high half broadcast) */
        if (MODE == 8) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n s_nop %4\n v_pk_fma_f32 %0, %1, %3, %0 op_sel_hi:[1,0,1]\n s_nop %4" : "+v"(x) : "s"(ks), "v"(c), "v"(cn), "n"(W)); }      /* ... low half broadcast */
        if (MODE == 9) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]\n s_nop %3\n v_pk_mul_f32 %0, %0, %2 op_sel:[0,0] op_sel_hi:[1,0]\n s_nop %3" : "+v"(x) : "v"(ka), "v"(kb), "n"(W)); }      /* broadcasts of a vector-register operand, not crossed */
        if (MODE == 10) { asm volatile("v_pk_mul_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n s_nop %3\n v_pk_mul_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n s_nop %3" : "+v"(x) : "v"(ka), "v"(kb), "n"(W)); }      /* sign modifiers only (the ORB descriptor kernel's rotation): no half select */
        if (MODE == 6) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]\n s_nop %3\n v_pk_mul_f32 %0, %0, %2 op_sel:[0,1] op_sel_hi:[1,0]\n s_nop %3" : "+v"(x) : "v"(ka), "v"(kb), "n"(W)); }
    }
    const uint32_t ux = __float_as_uint(x[0]), uy = __float_as_uint(x[1]);
    const bool differs = ux != (uint32_t)__builtin_amdgcn_readfirstlane((int)ux) || uy != (uint32_t)__builtin_amdgcn_readfirstlane((int)uy);
    if (differs) atomicAdd(&bad[lane >> 4], 1u);
    if (__any(differs) && lane == 0) atomicAdd(&bad[4], 1u);
    if (x[0] == 12345.678f) sink[0] = x[1] + pad;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int corun = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 50, vb = argc > 3 ? atoi(argv[3]) : 2048, iters = argc > 4 ? atoi(argv[4]) : 4000, rows = argc > 5 ? atoi(argv[5]) : 15, mode = argc > 6 ? atoi(argv[6]) : 0, wait = argc > 7 ? atoi(argv[7]) : 0;
    hipStream_t sV, sC; CK(hipStreamCreateWithFlags(&sV, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));
    int h_in[1024]; srand(7);
    for (int i = 0; i < 256; i++) { h_in[4 * i] = rand() & 0x3fffff; h_in[4 * i + 1] = (rand() & 0x3ff) - 512; h_in[4 * i + 2] = rand() & 0x3fffff; h_in[4 * i + 3] = (rand() & 0x3ff) - 512; }
    int *d_in; unsigned *d_bad; float *d_o, *d_sink;
    CK(hipMalloc(&d_in, sizeof h_in)); CK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice)); CK(hipMalloc(&d_bad, 32)); CK(hipMemset(d_bad, 0, 32)); CK(hipMalloc(&d_o, 4096 * 256 * 4)); CK(hipMalloc(&d_sink, 4));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; r++) {
        if (corun) for (int l = 0; l < 3; l++) hipLaunchKernelGGL(k_corun, dim3(1024), dim3(256), 0, sC, d_o, 2000, corun);
        const float k1 = 1.0009765625f, k2 = 0.9990234375f;
#define FORM(M, W) hipLaunchKernelGGL((k_form<M, W>), dim3(vb), dim3(256), 0, sV, iters, k1, k2, d_bad, d_sink)
        switch (mode * 10 + wait) {
        case 0: hipLaunchKernelGGL(k_victim, dim3(vb), dim3(256), 0, sV, d_in, iters, 812.5f, -37.25f, 640.75f, 1.f / (812.5f * 640.75f - 37.25f * 37.25f), d_bad, d_sink, rows); break;
        case 10: FORM(1, 0); break; case 11: FORM(1, 1); break; case 13: FORM(1, 3); break; case 17: FORM(1, 7); break;
        case 20: FORM(2, 0); break; case 21: FORM(2, 1); break; case 23: FORM(2, 3); break;
        case 30: FORM(3, 0); break; case 31: FORM(3, 1); break; case 33: FORM(3, 3); break;
        case 40: FORM(4, 0); break; case 50: FORM(5, 0); break;
        case 70: FORM(7, 0); break; case 80: FORM(8, 0); break; case 90: FORM(9, 0); break;
        case 100: FORM(10, 0); break;
        case 60: FORM(6, 0); break; case 61: FORM(6, 1); break; case 63: FORM(6, 3); break;
        default: printf("no such mode / wait\n"); return 1;
        }
        CK(hipDeviceSynchronize());
    }
    unsigned h_bad[8]; CK(hipMemcpy(h_bad, d_bad, 32, hipMemcpyDeviceToHost));
    printf("mode %d s_nop %d rows %d, co-runner %s, %d victim launches x %d workgroups x %d chain steps: waves with a lane that differs from lane 0: %u of %ld; differing lanes by quarter [0-15 16-31 32-47 48-63] = %u %u %u %u\n",
           mode, wait, rows, corun == 1 ? "bf16 matrix products" : corun == 2 ? "fp32 FMAs" : "none", reps, vb, iters, h_bad[4], (long)reps * vb * 4, h_bad[0], h_bad[1], h_bad[2], h_bad[3]);
    return 0;
}
