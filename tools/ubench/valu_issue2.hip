// Which VALU opcodes run at the 2-cycle rate and which at the 4-cycle rate (wave64, whole chip, 8 waves per SIMD, 4 independent chains per wave)?
// hipcc --offload-arch=gfx950 -O3 valu_issue2.hip -o valu_issue2 && ./valu_issue2
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OPS(X) \
  X(0, "v_and_b32 %0, %0, %1") X(1, "v_or_b32 %0, %0, %1") X(2, "v_xor_b32 %0, %0, %1") X(3, "v_lshlrev_b32 %0, 3, %0") X(4, "v_lshrrev_b32 %0, 3, %0") \
  X(5, "v_sub_u32 %0, %0, %1") X(6, "v_max_u32 %0, %0, %1") X(7, "v_min_i32 %0, %0, %1") X(8, "v_mov_b32 %0, %1") X(9, "v_add3_u32 %0, %0, %1, %2") \
  X(10, "v_and_or_b32 %0, %0, %1, %2") X(11, "v_mul_u32_u24 %0, %0, %1") X(12, "v_mad_u32_u24 %0, %0, %1, %2") X(13, "v_mul_f32 %0, %0, %1") X(14, "v_add_f32 %0, %0, %1") \
  X(15, "v_cvt_f32_i32 %0, %0") X(16, "v_mul_i32_i24 %0, %0, %1") X(17, "v_cmp_lt_u32 vcc, %0, %1") X(18, "v_cndmask_b32 %0, %0, %1, vcc") X(19, "v_ashrrev_i32 %0, 3, %0") \
  X(20, "v_lshl_add_u32 %0, %0, 2, %1") X(21, "v_add_lshl_u32 %0, %0, %1, 2") X(22, "v_pk_sub_i16 %0, %0, %1") X(23, "v_pk_max_u16 %0, %0, %1") X(24, "v_pk_min_u16 %0, %0, %1") \
  X(25, "v_pk_lshrrev_b16 %0, 1, %0") X(26, "v_pk_mul_lo_u16 %0, %0, %1") X(27, "v_max3_u32 %0, %0, %1, %2") X(28, "v_med3_i32 %0, %0, %1, %2") X(29, "v_bfi_b32 %0, %1, %0, %2") \
  X(30, "v_sub_u16 %0, %0, %1") X(31, "v_add_u16 %0, %0, %1") X(32, "v_max_u16 %0, %0, %1") X(33, "v_cmp_gt_u16 vcc, %0, %1") X(34, "v_bcnt_u32_b32 %0, %0, %1") \
  X(35, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1") X(36, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2") \
  X(37, "v_max_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3") X(38, "v_readlane_b32 s20, %0, 3") X(39, "v_rndne_f32 %0, %0") \
  X(40, "v_floor_f32 %0, %0") X(41, "v_cvt_i32_f32 %0, %0") X(42, "v_mbcnt_lo_u32_b32 %0, %1, %0") X(43, "v_ffbh_u32 %0, %0") X(44, "v_msad_u8 %0, %0, %1, %2") X(45, "v_lerp_u8 %0, %0, %1, %2") \
  X(46, "v_xad_u32 %0, %0, %1, %2") X(47, "v_or3_b32 %0, %0, %1, %2") X(48, "v_sub_f32 %0, %0, %1") X(49, "v_max_f32 %0, %0, %1") X(51, "v_fmac_f32 %0, %1, %2")
template <int OP>
__global__ void __launch_bounds__(256) k(unsigned *out, int iters, unsigned a, unsigned b)
{
    unsigned x[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
#define X(ID, STR) if (OP == ID) asm volatile(STR : "+v"(x[i]) : "v"(a), "v"(b) : "vcc", "s20");
                OPS(X)
#undef X
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x[0] + x[1] + x[2] + x[3];
}
template <int OP> void run(const char *name)
{
    unsigned *out; const int blocks = 256 * 8 * 4, iters = 1500;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 0x00030201u, 0x01000302u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, iters, 0x00030201u, 0x01000302u);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per = iters * 64.0 * blocks * 4 / (ms * 1e6) / 1024.0;
    printf("%5.2f cycles @2.4GHz   %s\n", 2.4 / per, name);
    hipFree(out);
}
int main()
{
#define X(ID, STR) run<ID>(STR);
    OPS(X)
#undef X
    return 0;
}
