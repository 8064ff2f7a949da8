// What does the fp32 -> 3 x bf16 operand split cost on gfx950?  Issue cost (cycles per wave instruction) of v_cvt_pk_bf16_f32 against plain VALU ops, and of the whole
// eight-value split (sgx_split3x8: 12 cvt + 16 unpack + 8 v_pk_add_f32) against a truncating variant built from v_and / v_pk_add / v_perm only, at 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off bf16_split.hip -o bf16_split && ./bf16_split
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float x, float y) { f2 v = {x, y}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2)); }
__device__ __forceinline__ void split_rne(const float (&v)[8], u4 &t0, u4 &t1, u4 &t2)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float x = v[2 * j], y = v[2 * j + 1];
        const unsigned p0 = pk(x, y);
        const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
        const unsigned p1 = pk(rx, ry);
        t0[j] = p0; t1[j] = p1; t2[j] = pk(rx - __uint_as_float(p1 << 16), ry - __uint_as_float(p1 & 0xffff0000u));
    }
}
__device__ __forceinline__ void split_trunc(const float (&v)[8], u4 &t0, u4 &t1, u4 &t2)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float x = v[2 * j], y = v[2 * j + 1];
        const float rx = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u), ry = y - __uint_as_float(__float_as_uint(y) & 0xffff0000u);
        const float sx = rx - __uint_as_float(__float_as_uint(rx) & 0xffff0000u), sy = ry - __uint_as_float(__float_as_uint(ry) & 0xffff0000u);
        t0[j] = __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u);
        t1[j] = __builtin_amdgcn_perm(__float_as_uint(ry), __float_as_uint(rx), 0x07060302u);
        t2[j] = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
    }
}
template <int MODE>
__global__ void k(float *out, int iters, long long *cyc)
{
    float v[8];
    for (int j = 0; j < 8; j++) v[j] = 1.0f + threadIdx.x * 1e-3f + j * 0.37f;
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {            // 8 independent v_cvt_pk_bf16_f32
            unsigned r[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r[j]) : "v"(v[j]), "v"(v[(j + 1) & 7])); }
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= r[j];
        } else if (MODE == 1) {     // 8 independent v_and_b32 (reference price of a plain VALU op)
            unsigned r[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { asm volatile("v_and_b32 %0, %1, %2" : "=v"(r[j]) : "v"(v[j]), "v"(v[(j + 1) & 7])); }
#pragma unroll
            for (int j = 0; j < 8; j++) acc ^= r[j];
        } else if (MODE == 2) { u4 a, b, c; split_rne(v, a, b, c); acc ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3]; }
        else { u4 a, b, c; split_trunc(v, a, b, c); acc ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3]; }
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] += __uint_as_float((acc & 1u) | 0x33000000u);      // keeps the loop's inputs changing (8 cheap ops)
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(acc) + v[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char *name, int threads)
{
    float *out; long long *cyc, h = 0; const int iters = 20000;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %4d threads per CU (%d waves per SIMD): %7.1f clock64 ticks per iteration\n", name, threads, threads / 256, (double)h / iters);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int th : {256, 512, 1024}) {
        run<0>("8 x v_cvt_pk_bf16_f32 + 16", th); run<1>("8 x v_and_b32 + 16", th);
        run<2>("split3x8 RNE (cvt) + 20", th); run<3>("split3x8 trunc (and/perm) + 20", th);
    }
    return 0;
}
