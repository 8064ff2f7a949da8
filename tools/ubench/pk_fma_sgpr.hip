// What does v_pk_fma_f32 read when a source is an SGPR pair?  (LLVM never emits that form: it copies scalar weights into VGPRs first.)
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/pk_fma_sgpr.hip -o /tmp/pk_fma_sgpr && /tmp/pk_fma_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *w, float *out)
{
    const float w0 = w[0], w1 = w[1];                       // uniform -> s_load
    f2 x; x.x = 10.f + threadIdx.x; x.y = 20.f + threadIdx.x;
    f2 ws; ws.x = w0; ws.y = w1;
    f2 c0 = {0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c0) : "s"(ws), "v"(x));                                        // plain
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(c1) : "s"(ws), "v"(x));                      // low half for both lanes
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c2) : "s"(ws), "v"(x));       // high half for both lanes
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c3) : "v"(ws), "v"(x));                                        // VGPR reference
    if (threadIdx.x == 1) { out[0] = c0.x; out[1] = c0.y; out[2] = c1.x; out[3] = c1.y; out[4] = c2.x; out[5] = c2.y; out[6] = c3.x; out[7] = c3.y; }
}
int main()
{
    float hw[2] = {3.f, 5.f}, *dw, *dout, ho[8];
    hipMalloc(&dw, 8); hipMalloc(&dout, 32); hipMemcpy(dw, hw, 8, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dw, dout); hipMemcpy(ho, dout, 32, hipMemcpyDeviceToHost);
    printf("x = (11, 21), w = (3, 5)\n");
    printf("sgpr plain        : %g %g   (pair semantics: 33 105)\n", ho[0], ho[1]);
    printf("sgpr op_sel_hi 0  : %g %g   (broadcast low: 33 63)\n", ho[2], ho[3]);
    printf("sgpr op_sel 1,hi 1: %g %g   (broadcast high: 55 105)\n", ho[4], ho[5]);
    printf("vgpr plain        : %g %g\n", ho[6], ho[7]);
    return 0;
}
