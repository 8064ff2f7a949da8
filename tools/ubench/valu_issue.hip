// Micro-benchmark: SIMD issue cost (cycles per wave64 instruction per SIMD) of the VALU operations the integer / byte kernels of this library are made of,
// with 8 waves per SIMD and 4 independent chains per wave (throughput regime).  One workgroup of 2048 threads on one CU; clock64() of wave 0.
// hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { OP_ADD_U32, OP_MAD_I24, OP_MUL_LO, OP_PERM, OP_DOT2, OP_PK_ADD_U16, OP_PK_MAD_U16, OP_ALIGNBIT, OP_FMA_F32, OP_PK_FMA_F32, OP_DPP_ADD, OP_CNDMASK, OP_BFE, OP_LSHL_OR, OP_FMA_F64, OP_ADD_F64, OP_SAD_U8, OP_MIN3, OP_COUNT };
static const char *names[] = { "v_add_u32", "v_mad_i32_i24", "v_mul_lo_u32", "v_perm_b32", "v_dot2c_i32_i16", "v_pk_add_u16", "v_pk_mad_u16", "v_alignbit_b32", "v_fma_f32", "v_pk_fma_f32",
                               "v_add_u32_dpp", "v_cndmask_b32", "v_bfe_u32", "v_lshl_or_b32", "v_fma_f64", "v_add_f64", "v_sad_u8", "v_min3_u32" };
template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned *out, unsigned long long *cyc, int iters, unsigned a, unsigned b)
{
    unsigned x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = threadIdx.x * 2654435761u + i;
    double d[4] = { (double)threadIdx.x, 1.5, 2.5, 3.5 };
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (OP == OP_MAD_I24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == OP_DOT2) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == OP_PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (OP == OP_PK_MAD_U16) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x[i]) : "v"(a));
                else if (OP == OP_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == OP_PK_FMA_F32) { unsigned long long *p = (unsigned long long *)&d[i]; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*p) : "v"(d[(i + 1) & 3])); }
                else if (OP == OP_DPP_ADD) asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
                else if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
                else if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(x[i]));
                else if (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x[i]) : "v"(a));
                else if (OP == OP_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 3] ));
                else if (OP == OP_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 3]));
                else if (OP == OP_SAD_U8) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                else if (OP == OP_MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            }
        }
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + (unsigned)d[0] + (unsigned)d[1] + (unsigned)d[2] + (unsigned)d[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> void run(int threads)
{
    // whole chip: 256 CUs x (2048 / threads) blocks, i.e. 8 waves per SIMD; wall time by HIP events -> wave instructions per ns per SIMD
    unsigned *out; unsigned long long *cyc, h;
    const int blocks = 256 * (2048 / threads) * 4;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0x00030201u, 0x01000302u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0x00030201u, 0x01000302u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double n_instr = iters * 16.0 * 4;                                   // per wave
    const double waves = (double)blocks * threads / 64.0;
    const double per_simd_per_ns = n_instr * waves / (ms * 1e6) / 1024.0;
    printf("%-18s %4d-thread blocks: %.3f wave-instructions / ns / SIMD = %.2f cycles per instruction at 2.4 GHz (clock64 view of wave 0: %.2f ticks per instruction)\n",
           names[OP], threads, per_simd_per_ns, 2.4 / per_simd_per_ns, (double)h / n_instr);
    hipFree(out); hipFree(cyc);
}
template <int OP> void both() { run<OP>(256); }
int main()
{
    both<OP_ADD_U32>(); both<OP_MAD_I24>(); both<OP_MUL_LO>(); both<OP_PERM>(); both<OP_DOT2>(); both<OP_PK_ADD_U16>(); both<OP_PK_MAD_U16>(); both<OP_ALIGNBIT>(); both<OP_FMA_F32>();
    both<OP_PK_FMA_F32>(); both<OP_DPP_ADD>(); both<OP_BFE>(); both<OP_LSHL_OR>(); both<OP_FMA_F64>(); both<OP_ADD_F64>(); both<OP_SAD_U8>(); both<OP_MIN3>();
    return 0;
}
