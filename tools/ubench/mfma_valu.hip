// Do fp32-input MFMAs (v_mfma_f32_32x32x2_f32) and plain fp32 VALU work of OTHER waves on the same SIMD overlap on gfx950?
// One workgroup of 512 threads per CU (2 waves per SIMD): waves 0-3 run an MFMA-only loop, waves 4-7 a v_fma_f32-only loop (mode 3), or only one of the two
// kinds does work (modes 1 / 2).  Overlap => t(both) ~ max(t1, t2); serialisation => t(both) ~ t1 + t2.  Also bf16 MFMA beside the same VALU loop for contrast.
// hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu && ./mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void __launch_bounds__(512) k(float *out, int iters, int mode, int valu_per)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            const float x = threadIdx.x * 1e-3f, y = 1.0001f;
            s8v xb = {1, 2, 3, 4, 5, 6, 7, 8}, yb = {1, 1, 1, 1, 1, 1, 1, 1};
            for (int it = 0; it < iters; it++) {
                if (KIND == 0) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a3, 0, 0, 0);
                }
            }
            for (int i = 0; i < 16; i++) r += a0[i] + a1[i] + a2[i] + a3[i];
        }
    } else {
        if (mode & 2) {
            float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f; const float m = 1.000001f, c = 1e-7f;
            for (int it = 0; it < iters * valu_per; it++) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(m), "v"(c));
            }
            r = v0 + v1 + v2 + v3;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND> float run(int mode, int valu_per)
{
    float *out; const int blocks = 256, iters = 20000;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(512), 0, 0, out, iters, mode, valu_per);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(512), 0, 0, out, iters, mode, valu_per);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); hipFree(out);
    return ms;
}
int main()
{
    // valu_per chosen so that the VALU loop alone takes about as long as the MFMA loop alone: 4 MFMAs = 256 cycles (f32) ~ 4 x 16 v_fma at 4 cycles
    for (int vp : {8, 16, 32}) {
        const float a = run<0>(1, vp), b = run<0>(2, vp), c = run<0>(3, vp);
        printf("f32  MFMA 32x32x2 : valu_per %2d  mfma-only %.3f ms  valu-only %.3f ms  both %.3f ms  (max %.3f, sum %.3f)\n", vp, a, b, c, a > b ? a : b, a + b);
    }
    for (int vp : {1, 2, 4}) {
        const float a = run<1>(1, vp), b = run<1>(2, vp), c = run<1>(3, vp);
        printf("bf16 MFMA 32x32x16: valu_per %2d  mfma-only %.3f ms  valu-only %.3f ms  both %.3f ms  (max %.3f, sum %.3f)\n", vp, a, b, c, a > b ? a : b, a + b);
    }
    return 0;
}
