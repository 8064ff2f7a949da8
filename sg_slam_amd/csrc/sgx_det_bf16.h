// sgx_det_bf16.h — the detector's 1x1 convolutions on the bf16 matrix pipes with fp32 accuracy ("bf16x3").
//
// Why.  On gfx950 the fp32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate and occupies the vector issue path: it does not overlap with the VALU work of
// other waves (tools/ubench/mfma_valu.hip, profiles/r3_ubench_mfma_valu.txt), so a fused block pays "MFMA cycles + VALU cycles".  v_mfma_f32_32x32x16_bf16 does 8x the
// multiply-adds per instruction in half the cycles (16x the rate) and DOES run beside the vector work of other waves.
//
// Arithmetic.  An fp32 value x is split exactly into three bf16 terms, x = x0 + x1 + x2 with x0 = RN_bf16(x), x1 = RN_bf16(x - x0), x2 = x - x0 - x1 (both differences are
// exact in fp32; |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|, and x2 has at most 8 significant bits, so it IS a bf16).  A product a*b is evaluated as the six leading cross terms
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)
// each of them exact in the fp32 accumulator's input (8 x 8-bit significands); the three dropped terms are <= 3 * 2^-24 |a b|: the product error of an fp32 multiply-add.
// Accumulation is fp32 inside the MFMA, 16 k per instruction.  This is NOT the ascending-k fmaf chain of k_conv_pw2 / the oracle: results differ from the exact-fp32 plan
// in the last bits, like any other fp32 summation order (and like the reference's own ncnn, which runs neither — Detector2D.cc:22 sets use_vulkan_compute, the CPU path
// uses packed SIMD layouts).  Parity criterion: tests/test_detector.py::run_compare (drift against the oracle's float64 run within 4x of the oracle's own float32 drift,
// DetectionOutput rows identical); the exact-fp32 plan stays selectable (SGX_DET_GEMM=f32, sgx_det_debug_set_gemm(0)) and is the anchor of the plan-equality tests.
//
// Operand layouts of v_mfma_f32_32x32x16_bf16 (lane l: half = l >> 5, i = l & 31): A[i][k], B[k][i] with k = 8 half + j, j = 0..7 = the eight bf16 of the lane's four
// operand registers; C/D as the fp32 forms (row = (r & 3) + 8 (r >> 2) + 4 half, column i).  Weights are split on the host into
//     Ws[k16 step][term 0..2][half][oc (padded to ldw)][8 bf16]          (16 bytes per lane and term: one dwordx4 load is one A operand)
// with zero rows past the layer's input channels; activations are split in registers.
#pragma once
#include "sgx_det_kernels.h"
#include <string.h>

#ifndef SGX_EMU
typedef unsigned sgx_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sgx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sgx_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sgx_f32x2 __attribute__((ext_vector_type(2)));

// two floats -> packed bf16 pair (x in the low half), round to nearest even: one v_cvt_pk_bf16_f32
SGX_DEV unsigned sgx_pk_bf16(float x, float y) { const sgx_f32x2 v = { x, y }; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sgx_bf16x2)); }

// exact three-term split of a pair: 3 conversions + 4 unpacks + 2 packed subtractions
SGX_DEV void sgx_split3(float x, float y, unsigned &p0, unsigned &p1, unsigned &p2)
{
    p0 = sgx_pk_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = sgx_pk_bf16(rx, ry);
    p2 = sgx_pk_bf16(rx - __uint_as_float(p1 << 16), ry - __uint_as_float(p1 & 0xffff0000u));
}

// the eight fp32 values of a lane's B (or A) operand -> the three bf16 operands
struct SgxB3 { sgx_u32x4 t0, t1, t2; };
SGX_DEV SgxB3 sgx_split3x8(const float (&v)[8])
{
    SgxB3 b;
#pragma unroll
    for (int j = 0; j < 4; j++) { unsigned p, q, r; sgx_split3(v[2 * j], v[2 * j + 1], p, q, r); b.t0[j] = p; b.t1[j] = q; b.t2[j] = r; }
    return b;
}

#define SGX_MFMA_BF16(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sgx_bf16x8, a_), __builtin_bit_cast(sgx_bf16x8, b_), c_, 0, 0, 0)
// acc += A x B over 16 k with fp32 accuracy: the three small cross terms first, the leading one last
SGX_DEV sgx_f32x16 sgx_mfma_bf16x3(const sgx_u32x4 &a0, const sgx_u32x4 &a1, const sgx_u32x4 &a2, const SgxB3 &b, sgx_f32x16 acc)
{
    acc = SGX_MFMA_BF16(a0, b.t2, acc); acc = SGX_MFMA_BF16(a1, b.t1, acc); acc = SGX_MFMA_BF16(a2, b.t0, acc);
    acc = SGX_MFMA_BF16(a0, b.t1, acc); acc = SGX_MFMA_BF16(a1, b.t0, acc);
    return SGX_MFMA_BF16(a0, b.t0, acc);
}

// ---------------------------------------------------------------------------------------------
// k_conv_pw3: the pointwise convolution of k_conv_pw2 (same work decomposition, grid, bias-initialised accumulators and epilogue) with the products on
// v_mfma_f32_32x32x16_bf16 (bf16x3).  Per k16 step a lane loads its eight input channels (k = 16 s + 8 half + j; 32 lanes along pixels: two rows of 128 B per load),
// splits them once and multiplies them with OCB weight tiles; weights come straight from global memory (L2-resident, one dwordx4 per term and tile).  B operands are
// requested two k16 steps ahead, A operands one.  grid / XCD order / tile shapes as k_conv_pw2.
// ---------------------------------------------------------------------------------------------
#ifndef SGX_PW3_BRING
#define SGX_PW3_BRING 3      /* B-operand ring: k16 steps in flight (A/B taps, tools/ab_build.sh) */
#endif
#ifndef SGX_PW3_ARING
#define SGX_PW3_ARING 2      /* A-operand ring */
#endif
#ifndef SGX_PW3_OCC4
#define SGX_PW3_OCC4 2       /* waves per SIMD the four-tile shapes are compiled for */
#endif
template <int OCB, int PXB>
#if !defined(SGX_PW3_V1)
#define SGX_PW3_OCC(OCB_, PXB_) ((OCB_) * (PXB_) == 1 ? 4 : ((PXB_) == 1 || (OCB_) * (PXB_) <= 3 ? 3 : 2))      /* with the weights in LDS every single-pixel-tile shape fits three waves per SIMD */
#else
#define SGX_PW3_OCC(OCB_, PXB_) ((OCB_) * (PXB_) == 1 ? 4 : ((OCB_) * (PXB_) <= 3 ? 3 : ((OCB_) * (PXB_) == 4 ? SGX_PW3_OCC4 : 2)))
#endif
SGX_KERNEL_OCC(256, SGX_PW3_OCC(OCB, PXB)) k_conv_pw3(int inc, int outc, int N, int total, const float *in, size_t in_pitch, const sgx_u32x4 *__restrict__ Ws, const float *bias,
                                                          float *out, size_t out_pitch, SgxEpi epi, int hwc, int hwc_off, int nxt, int noc, int ldw, int direct)
{
    constexpr int OCT = 32 * OCB;
    SGX_LDS float Es[4][32][33];                    // per-wave epilogue staging tile
    SGX_LDS float Bs[OCT];                          // bias of the oc block
    const int id = (int)blockIdx.x;
    const int grp = id / (8 * noc), rem = id - grp * (8 * noc);
    const int xt = grp * 8 + (rem & 7), yt = rem >> 3;                  // XCD = id % 8 = xt % 8
    if (xt >= nxt) return;                                               // uniform per workgroup
    const int oc0 = yt * OCT;
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int g0 = (xt * 4 + wave) * (32 * PXB);
    // lane offsets (bytes): image + pixel (+ the half-wave's row offset for the stores, as k_conv_pw2); the B loads add 8 half rows
    unsigned ioff4[PXB], ooff4[PXB], toff4[PXB];
#pragma unroll
    for (int m = 0; m < PXB; m++) {
        const unsigned gg = (unsigned)min(g0 + 32 * m + l31, total - 1), b = gg / (unsigned)N, n = gg - b * (unsigned)N, hn = n + (unsigned)half * (unsigned)N;
        ioff4[m] = (b * (unsigned)in_pitch + n) * 4u; ooff4[m] = (b * (unsigned)out_pitch + hn) * 4u; toff4[m] = (b * (unsigned)epi.tpitch + hn) * 4u;
    }
    if (tid < OCT) Bs[tid] = bias[min(oc0 + tid, outc - 1)];
    const int nks = (inc + 15) >> 4;
    const unsigned rowb = (unsigned)N * 4u;
    // B operand of k16 step s: rows 16 s + 8 half + j.  Rows past the last input channel (last step of a layer whose channel count is not a multiple of 16) are clamped
    // per lane — they meet zero weights; all other steps use a wave-uniform row base + the lane's offset
    auto loadB = [&](int s, float (&dst)[PXB][8]) {
        const int k0 = 16 * s;
        if (k0 + 16 <= inc) {
            const unsigned h8 = (unsigned)(8 * half) * rowb;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float *rowp = (const float *)((const char *)in + (unsigned)(k0 + j) * rowb);
#pragma unroll
                for (int m = 0; m < PXB; m++) dst[m][j] = sgx_ldoff(rowp, ioff4[m] + h8);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned roff = (unsigned)min(k0 + 8 * half + j, inc - 1) * rowb;
#pragma unroll
                for (int m = 0; m < PXB; m++) dst[m][j] = sgx_ldoff(in, ioff4[m] + roff);
            }
        }
    };
    const sgx_u32x4 *wl = Ws + (size_t)half * ldw + oc0 + l31;          // + (6 s + 2 term) ldw + 32 tile
    auto loadA = [&](int s, sgx_u32x4 (&dst)[OCB][3]) {
        const sgx_u32x4 *ws = wl + (size_t)(6 * s) * ldw;
#pragma unroll
        for (int t = 0; t < OCB; t++)
#pragma unroll
            for (int q = 0; q < 3; q++) dst[t][q] = ws[(size_t)(2 * q) * ldw + 32 * t];
    };
#if !defined(SGX_PW3_V1)
    // Round 4, second version: the weights of a k16 step go through LDS ONCE per workgroup (its four waves multiply different pixels by the same weights) instead of once per wave
    // from L2: a quarter of the weight loads, and 12 operand registers per wave instead of the 24 OCB of a private double buffer — the (5,1) shape drops from 245 to ~150
    // registers, i.e. from two to three waves per SIMD, on layers whose grids (10 x 10 and 19 x 19 maps) give a SIMD only two or three waves to hide a k loop of 10-60 steps behind.
    // Double-buffered chunk of one k16 step, [term * 2 + half][oc] x 16 bytes = the global layout cut to the block's oc range; one barrier per step.
    constexpr int BR = 3, ROWS = 6 * OCT, CP = (ROWS + 255) / 256;
    SGX_LDS sgx_u32x4 Aw[2][ROWS];
    float braw[BR][PXB][8];
    sgx_u32x4 cpy[CP];
    int cidx[CP]; unsigned coff[CP];                                   // this thread's slots of the chunk copy: LDS index and global offset inside a step (in 16-byte units)
#pragma unroll
    for (int i = 0; i < CP; i++) { const int idx = tid + 256 * i, row = idx / OCT, oc = idx - row * OCT; cidx[i] = idx; coff[i] = (unsigned)(row * ldw + oc0 + oc); }
    auto fetchA = [&](int s_) {
        const sgx_u32x4 *ws = Ws + (size_t)(6 * s_) * ldw;
#pragma unroll
        for (int i = 0; i < CP; i++) if (CP * 256 == ROWS || cidx[i] < ROWS) cpy[i] = ws[coff[i]];
    };
    auto storeA = [&](int buf_) {
#pragma unroll
        for (int i = 0; i < CP; i++) if (CP * 256 == ROWS || cidx[i] < ROWS) Aw[buf_][cidx[i]] = cpy[i];
    };
    fetchA(0); loadB(0, braw[0]); loadB(min(1, nks - 1), braw[1]);
    storeA(0);
    fetchA(min(1, nks - 1));
    __syncthreads();
    sgx_f32x16 acc[OCB][PXB];
#pragma unroll
    for (int t = 0; t < OCB; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float bz = Bs[32 * t + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
            for (int m = 0; m < PXB; m++) acc[t][m][r] = bz;
        }
    for (int s0 = 0; s0 < nks; s0 += 6) {
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const int s_ = s0 + d;
            if (s_ < nks) {                                              // uniform
                const int buf = s_ & 1;
                loadB(min(s_ + 2, nks - 1), braw[(d + 2) % 3]);
                SgxB3 bs[PXB];
#pragma unroll
                for (int m = 0; m < PXB; m++) bs[m] = sgx_split3x8(braw[d % 3][m]);
                const sgx_u32x4 *al = &Aw[buf][half * OCT + l31];
                // (issuing one cross term for every tile before the next term — OCB * PXB independent MFMAs between two dependent ones — was measured: equal on every shape
                // but (5,1), where it spills and runs three times slower)
#pragma unroll
                for (int t = 0; t < OCB; t++) {
                    const sgx_u32x4 a0 = al[32 * t], a1 = al[2 * OCT + 32 * t], a2 = al[4 * OCT + 32 * t];
#pragma unroll
                    for (int m = 0; m < PXB; m++) acc[t][m] = sgx_mfma_bf16x3(a0, a1, a2, bs[m], acc[t][m]);
                }
                storeA(buf ^ 1);                                         // the chunk of step s + 1 (in registers since the last step) -> the buffer read during step s - 1
                fetchA(min(s_ + 2, nks - 1));
                __syncthreads();
            }
        }
    }
#else
    constexpr int BR = SGX_PW3_BRING, AR = SGX_PW3_ARING;
    float braw[BR][PXB][8];
    sgx_u32x4 aw[AR][OCB][3];
    loadB(0, braw[0]); loadA(0, aw[0]);
    if (BR == 3) loadB(min(1, nks - 1), braw[1]);
    __syncthreads();
    sgx_f32x16 acc[OCB][PXB];
#pragma unroll
    for (int t = 0; t < OCB; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float bz = Bs[32 * t + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
            for (int m = 0; m < PXB; m++) acc[t][m][r] = bz;
        }
    // six steps per trip so that the ring slots (3 for B, 2 for A) are compile-time indices
    for (int s0 = 0; s0 < nks; s0 += 6) {
#pragma unroll
        for (int d = 0; d < 6; d++) {
            const int s = s0 + d;
            if (s < nks) {                                                // uniform
                loadB(min(s + BR - 1, nks - 1), braw[(d + BR - 1) % BR]);
                if (AR == 2) loadA(min(s + 1, nks - 1), aw[(d + 1) & 1]);
                SgxB3 bs[PXB];
#pragma unroll
                for (int m = 0; m < PXB; m++) bs[m] = sgx_split3x8(braw[d % BR][m]);
#pragma unroll
                for (int t = 0; t < OCB; t++)
#pragma unroll
                    for (int m = 0; m < PXB; m++) acc[t][m] = sgx_mfma_bf16x3(aw[d % AR][t][0], aw[d % AR][t][1], aw[d % AR][t][2], bs[m], acc[t][m]);
                if (AR == 1) loadA(min(s + 1, nks - 1), aw[0]);          // single set: the next step's weights are requested behind this step's products
            }
        }
    }
#endif
    sgx_pw2_epilogue<OCB, PXB>(epi, acc, Es[wave], oc0, outc, N, total, g0, half, l31, out, out_pitch, ooff4, toff4, hwc, hwc_off, direct);
}
#endif

// host side: [oc][K] fp32 weights (ncnn order) -> Ws[ceil(K / 16)][3][2][ldw][8] bf16 as 16-bit patterns, zero beyond K and outc
static inline unsigned short sgx_bf16_rne(float x)
{
    unsigned u; memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);           // inf / NaN: truncate
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float sgx_bf16_to_f32(unsigned short h) { const unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; }
static inline void sgx_split_weights_bf16x3(const float *w, int outc, int K, int ldw, unsigned short *dst /* ceil(K/16) * 6 * ldw * 8 */)
{
    const int nks = (K + 15) / 16;
    memset(dst, 0, (size_t)nks * 6 * ldw * 8 * sizeof(unsigned short));
    for (int o = 0; o < outc; o++)
        for (int k = 0; k < K; k++) {
            const float x = w[(size_t)o * K + k];
            const unsigned short h0 = sgx_bf16_rne(x); const float r1 = x - sgx_bf16_to_f32(h0);
            const unsigned short h1 = sgx_bf16_rne(r1); const float r2 = r1 - sgx_bf16_to_f32(h1);
            const unsigned short h2 = sgx_bf16_rne(r2);
            const int s = k >> 4, hf = (k >> 3) & 1, j = k & 7;
            const unsigned short hs[3] = { h0, h1, h2 };
            for (int t = 0; t < 3; t++) dst[((((size_t)s * 3 + t) * 2 + hf) * ldw + o) * 8 + j] = hs[t];
        }
}
