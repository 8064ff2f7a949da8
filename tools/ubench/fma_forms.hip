// Micro-benchmark: issue cost of the fp32 FMA forms the fused-block kernels can be built from, as a function of occupancy (waves per SIMD) and of the number of
// independent accumulator chains per wave:  v_fma_f32 (VGPR operands), v_fmac_f32 with the multiplier in an SGPR, v_pk_fma_f32 (VGPR pairs), v_pk_fma_f32 with an SGPR pair
// and op_sel broadcast (the form k_fused_block2 uses).  Whole chip, W workgroups of 256 threads per CU (W waves per SIMD); HIP-event time -> cycles per wave instruction per SIMD.
// hipcc --offload-arch=gfx950 -O3 fma_forms.hip -o fma_forms && ./fma_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { F_FMA_V, F_FMAC_S, F_PK_V, F_PK_S, F_COUNT };
static const char *names[] = { "v_fma_f32 v,v,v", "v_fmac_f32 v,s,v", "v_pk_fma_f32 v,v,v", "v_pk_fma_f32 v,s,v op_sel" };
template <int FORM, int CH>
__global__ void __launch_bounds__(256) k(float *out, int iters, float wa, float wb)
{
    f32x2 acc[CH], x; x.x = threadIdx.x * 1e-3f; x.y = 1.0f - x.x;
    f32x2 w; w.x = wa; w.y = wb;
#pragma unroll
    for (int i = 0; i < CH; i++) { acc[i].x = (float)i; acc[i].y = (float)(i + 1); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / CH; r++) {
#pragma unroll
            for (int i = 0; i < CH; i++) {
                if (FORM == F_FMA_V) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(x.x), "v"(x.y));
                else if (FORM == F_FMAC_S) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i].x) : "s"(wa), "v"(x.y));
                else if (FORM == F_PK_V) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(x));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"(w), "v"(x));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int FORM, int CH> double run(int W)
{
    float *out; const int blocks = 256 * W, iters = 4000;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FORM, CH>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<FORM, CH>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    const double n_instr = (double)iters * 64 * W;                 // wave instructions per SIMD (one wave of each workgroup per SIMD)
    return ms * 1e-3 * 2.4e9 / n_instr;
}
template <int FORM> void form()
{
    printf("%-28s  cycles per wave instruction per SIMD (2.4 GHz); rows: independent chains per wave 1 / 2 / 4 / 8, columns: 1 / 2 / 3 / 4 waves per SIMD\n", names[FORM]);
    printf("   chains 1: %6.2f %6.2f %6.2f %6.2f\n", run<FORM, 1>(1), run<FORM, 1>(2), run<FORM, 1>(3), run<FORM, 1>(4));
    printf("   chains 2: %6.2f %6.2f %6.2f %6.2f\n", run<FORM, 2>(1), run<FORM, 2>(2), run<FORM, 2>(3), run<FORM, 2>(4));
    printf("   chains 4: %6.2f %6.2f %6.2f %6.2f\n", run<FORM, 4>(1), run<FORM, 4>(2), run<FORM, 4>(3), run<FORM, 4>(4));
    printf("   chains 8: %6.2f %6.2f %6.2f %6.2f\n", run<FORM, 8>(1), run<FORM, 8>(2), run<FORM, 8>(3), run<FORM, 8>(4));
}
int main() { form<F_FMA_V>(); form<F_FMAC_S>(); form<F_PK_V>(); form<F_PK_S>(); return 0; }
