// sgx_det_irb.h — inverted-residual block of the detector backbone (MobileNetV3: ncnn graph mobilenetv3_ssdlite_voc.param, caller Detector2D.cc:34-45)
// as ONE kernel on the fp32 matrix cores:
//     [pointwise expand Cin -> Cexp + act]  ->  depthwise K x K stride S + act  ->  pointwise project Cexp -> Cout
//     [-> squeeze Cout -> Cq + ReLU -> excite Cq -> Cout -> hard-sigmoid gate x project output]  [+ residual]
// (this graph's squeeze-excite has NO pooling: it is a per-pixel 1x1 chain).  The SSD heads (depthwise 3x3 -> pointwise, HWC store) and blocks whose
// expanded tensor has other readers run the same kernel without the expand stage (the depthwise input is then read from HBM).
//
// Work decomposition.  A workgroup owns G whole images (small maps) or one band of output rows of one image; wave w owns output-pixel group w
// (32 pixels: the N side of v_mfma_f32_32x32x2_f32) for the WHOLE block, so everything behind the depthwise stage is wave-private:
//   expand   E[32 ch][in pixels] = act1(b1 + W1 x X): one 32 x 32 MFMA tile per (chunk of 32 expanded channels, input-pixel group), tiles dealt round-robin
//            to the waves; A = weights (host-transposed, zero-padded), B = input straight from global memory in the MFMA lane layout; the activated tile is
//            written to a zero-bordered LDS plane buffer (the depthwise convolution's zero padding) — the ONLY cross-wave hand-off (one barrier per chunk
//            with two plane buffers, two with one)
//   dw       lane (half h, pixel j) of wave w computes channel 2s + h of the chunk at ITS output pixel from the LDS planes (K*K taps, fmaf in tap order):
//            the result IS the B operand of k-step s of the project GEMM in the MFMA lane layout — the depthwise output never exists in memory
//   project  acc[t] (Cout / 32 tiles x 32 pixels, 16 registers each) += W2[:, 2s..2s+1] x that operand; accumulators persist over the chunks
//   squeeze-excite  the accumulator (C/D) layout is turned into the B layout with eight v_permlane32_swap per tile (rows 8m+{0..3} live in the lower,
//            8m+4+{0..3} in the upper half-wave: swapping register pairs across the halves yields the (k, k+1) row pairs in ascending k), so both
//            1x1 convolutions of the gate chain run on the matrix cores out of registers; gate, residual add and the store (CHW or HWC) finish per lane.
// Arithmetic: accumulators start from the bias, products in ascending k (the MFMA is an exact fp32 fmaf chain), depthwise taps (i, j) ascending with fmaf,
// elementwise programs as the per-layer kernels apply them — the block is bit-identical to the per-layer plan (tests: *_fused_equals_unfused).
#pragma once
#include "sgx_det_kernels.h"
#include "sgx_det_bf16.h"
#ifdef SGX_EMU
#include <vector>
#endif

struct SgxIrb {
    int Cin, Cexp, Cout, Cq;                    // Cq = 0: no squeeze-excite
    int H, W, Ho, Wo, K, S, pad;
    int G, nbands, OH, batch;                   // images per workgroup (nbands == 1) or bands of OH output rows per image (G == 1)
    int Wp, HpWp, planeT, nbuf;                 // LDS plane geometry in floats: row pitch, per-image plane, per-channel plane (G images); 1 or 2 plane buffers
    int has_expand, stagger;
    int w1rows;                                 // > 0: the expand weights of a chunk go through LDS once per workgroup ([2][w1rows][32] floats behind the depthwise weights; round 4), k_irb only
    int act1, act2;                             // SGX_EMODE_ACT / SGX_EMODE_HSWISH
    float a1c1, a1lo, a1hi, a1c2, a2c1, a2lo, a2hi, a2c2;
    float qlo, qhi;                             // squeeze activation
    float gc1, glo, ghi, gc2;                   // gate: clip(v + gc1, glo, ghi) / gc2
    int has_res, hwc, hwc_off;
    const float *in; size_t in_pitch; float *out; size_t out_pitch; const float *res; size_t res_pitch;
    const float *w1T, *b1; int ld1;             // [ceil32(Cin)][ld1]
    const float *wdp;                           // [ceil32(Cexp)][KKP]: K*K taps, bias, zero padding
    const float *w2T, *b2; int ld2;             // [ceil32(Cexp)][ld2]
    const float *wq1T, *bq1; int ldq1;          // [ceil32(Cout)][ldq1]
    const float *wq2T, *bq2; int ldq2;          // [ceil32(Cq)][ldq2]
    // second head on the same input (SSD loc + conf heads of one feature map: depthwise -> pointwise twice from ONE staging of the input planes); Cout2 = 0: none
    int Cout2, hwc_off2, ld2b; const float *wdp2, *w2Tb, *b2b; float *out2; size_t out2_pitch;
    // original ncnn layouts (emulator build)
    const float *w1, *wd, *bd, *w2, *wq1, *wq2, *wd_b, *bd_b, *w2_b;
    // bf16x3 plan (k_irb3, sgx_det_bf16.h): the same weights split into three bf16 terms in the operand layout of v_mfma_f32_32x32x16_bf16, [k16 step][term][half][ld][8];
    // ld = the ld of the fp32 copy.  gemm = 1 selects k_irb3
    int gemm, dbg;                              // dbg: timing taps of k_irb3 (SGX_IRB3_DBG, wrong results): 1 no depthwise arithmetic, 2 no stage-B MFMAs, 4 no stage-A MFMAs, 8 no stage-A split, 16 no stage B at all, 32 no stage A tiles
    const void *w1S, *w2S, *wq1S, *wq2S, *w2Sb;
    // experiment (round 6, tap build): the block input ALSO as three bf16 terms in B-operand layout [k16 step][term][half][ldS pixels][8] per image (k_presplit): the bf16x3 expand
    // stage then loads operands (three 16-byte loads per k16 step) instead of eight dwords + a 44-instruction split
    const void *inS; int ldS;
};
#define SGX_IRB_KKP(K) (((K) * (K) + 1 + 3) & ~3)
static inline size_t sgx_irb_lds_bytes(const SgxIrb &p) { return (size_t)p.nbuf * 32 * ((size_t)p.planeT + (p.Cout2 ? 2 : 1) * SGX_IRB_KKP(p.K)) * 4 + (size_t)2 * p.w1rows * 32 * 4; }

SGX_DEV float sgx_irb_act(int mode, float v, float c1, float lo, float hi, float c2)
{
    if (mode == SGX_EMODE_HSWISH) { float u = v + c1; u = sgx_clipf(u, lo, hi); u = u * v; return sgx_div_c2(u, c2); }
    return sgx_clipf(v, lo, hi);
}

#ifndef SGX_EMU
// accumulator layout (lane l: pixel l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)) -> B-operand layout of the 16 k-steps over the tile's 32 rows
// (k-step j: lower half-wave row 2j, upper half-wave row 2j + 1).  v_permlane32_swap(x, y): x.hi <-> y.lo.
SGX_DEV void sgx_irb_d2b(const sgx_f32x16 &d, float (&b)[16])
{
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const auto r01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[4 * m]), __float_as_uint(d[4 * m + 1]), false, false);
        const auto r23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[4 * m + 2]), __float_as_uint(d[4 * m + 3]), false, false);
        b[4 * m + 0] = __uint_as_float(r01[0]); b[4 * m + 1] = __uint_as_float(r23[0]);
        b[4 * m + 2] = __uint_as_float(r01[1]); b[4 * m + 3] = __uint_as_float(r23[1]);
    }
}
// operand streams are buffer loads: 128-bit descriptor (wave-uniform) + 32-bit lane byte offset + scalar byte offset: no per-load address arithmetic
typedef __amdgpu_buffer_rsrc_t sgx_rsrc;
SGX_DEV sgx_rsrc sgx_mkrsrc(const void *ptr) { return __builtin_amdgcn_make_buffer_rsrc((void *)ptr, 0, 0x7fffffff, 0x00020000); }
SGX_DEV sgx_rsrc sgx_mkrsrc_n(const void *ptr, int bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)ptr, 0, bytes, 0x00020000); }      // loads past `bytes` return 0
SGX_DEV float sgx_bld(sgx_rsrc r, unsigned voff, unsigned soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0)); }
SGX_DEV void sgx_bst(sgx_rsrc r, unsigned voff, unsigned soff, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)voff, (int)soff, 0); }
// bias of accumulator register r of 32-row tile t: row = 32 t + (r & 3) + 8 (r >> 2) + 4 half; the descriptor ends at the last channel, so padded rows read 0
#define SGX_IRB_BIAS(rs, t, r, half) sgx_bld(rs, (unsigned)(half) * 16u, (unsigned)(32 * (t) + ((r) & 3) + 8 * ((r) >> 2)) * 4u)

template <int K, int S, int NT, int NQ, bool EXPAND, bool HS, int NT2, bool A3 = false>
__global__ void __launch_bounds__(768) k_irb(SgxIrb p)
{
    extern __shared__ __attribute__((aligned(16))) float sgx_irb_smem[];
    constexpr int KK = K * K, KKP = SGX_IRB_KKP(K);
    constexpr int AMODE = HS ? SGX_EMODE_HSWISH : SGX_EMODE_ACT;      // both activations of a block are of one kind in this graph (planner checks)
    float *Eb = sgx_irb_smem;                                         // [nbuf][32][planeT]
    float *Wds = sgx_irb_smem + (size_t)p.nbuf * 32 * p.planeT;       // [nbuf][32][KKP]
    float *Wds2 = Wds + (size_t)p.nbuf * 32 * KKP;                    // second head's depthwise weights (NT2 > 0)
    const int tid = (int)threadIdx.x, nthreads = (int)blockDim.x, wave = tid >> 6, nwaves = nthreads >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W, HWo = p.Ho * p.Wo;
    int b0, nimg, oy0, OH;
    if (p.nbands > 1) { b0 = (int)blockIdx.x / p.nbands; nimg = 1; oy0 = ((int)blockIdx.x - b0 * p.nbands) * p.OH; OH = min(p.OH, p.Ho - oy0); }
    else { b0 = (int)blockIdx.x * p.G; nimg = min(p.G, p.batch - b0); oy0 = 0; OH = p.Ho; }
    const int ypA = oy0 * S;                                          // first padded input row held in the planes
    const int iyA = max(0, ypA - p.pad), iyB = min(p.H, (oy0 + OH - 1) * S + K - p.pad), IH = iyB - iyA;      // real input rows [iyA, iyB)
    const int PI = nimg * IH * p.W, PO = nimg * OH * p.Wo, ngi = (PI + 31) >> 5, ngo = (PO + 31) >> 5;
    // my output pixel (waves past the last output group only help with stage A)
    const bool owner = wave < ngo;
    const int og = wave * 32 + l31; const bool ovalid = og < PO;
    const int oc_ = min(og, PO - 1), og_img = oc_ / (OH * p.Wo), orem = oc_ - og_img * (OH * p.Wo), oyl = orem / p.Wo, ox = orem - oyl * p.Wo;
    const int e_r = half * p.planeT + og_img * p.HpWp + oyl * S * p.Wp + ox * S;                // depthwise read base (tap (i, j): + i Wp + j; k-step s: + 2 s planeT)
    const unsigned opix = (unsigned)((oy0 + oyl) * p.Wo + ox);
    const sgx_rsrc r_in = sgx_mkrsrc(p.in), r_w1 = sgx_mkrsrc(EXPAND ? p.w1T : p.w2T), r_w2 = sgx_mkrsrc(p.w2T), r_b1 = sgx_mkrsrc_n(EXPAND ? p.b1 : p.b2, (EXPAND ? p.Cexp : p.Cout) * 4);

    {   // zero the plane buffers once: borders and rows outside the image stay zero, the interior is rewritten per chunk
        float4 *z = (float4 *)Eb; const int nz = p.nbuf * 8 * p.planeT;
        for (int i = tid; i < nz; i += nthreads) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Expand weights through LDS (round 4): the twelve waves of the workgroup multiply different pixels by the SAME 32 x Cin weight slice of a chunk; it is copied global -> LDS once
    // per workgroup, one chunk ahead (two buffers), instead of being fetched by every wave from L2 (p.w1rows = 0: the per-wave loads of round 3)
    float *W1s = Wds + (size_t)p.nbuf * 32 * KKP * (NT2 > 0 ? 2 : 1);
    auto stageW1 = [&](int chunk) {
        float4 *dst = (float4 *)(W1s + (size_t)(chunk & 1) * p.w1rows * 32);
        for (int i = tid; i < p.w1rows * 8; i += nthreads) dst[i] = *(const float4 *)(p.w1T + (size_t)(i >> 3) * p.ld1 + chunk * 32 + 4 * (i & 7));
    };
    if (EXPAND && !A3 && p.w1rows) stageW1(0);
    __syncthreads();                                                  // the zero-fill (and the first weight slice) before any wave writes a tile into the planes
    sgx_f32x16 acc[NT];
    {
        const sgx_rsrc r_b2 = sgx_mkrsrc_n(p.b2, p.Cout * 4);
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = SGX_IRB_BIAS(r_b2, t, r, half);
    }
    constexpr int NT2A = NT2 > 0 ? NT2 : 1;
    sgx_f32x16 acc2[NT2A];
    if (NT2 > 0) {
        const sgx_rsrc r_b2b = sgx_mkrsrc_n(p.b2b, p.Cout2 * 4);
#pragma unroll
        for (int t = 0; t < NT2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[t][r] = SGX_IRB_BIAS(r_b2b, t, r, half);
    }
    const sgx_rsrc r_w2b = sgx_mkrsrc(NT2 > 0 ? p.w2Tb : p.w2T);
    const unsigned aoff2b = (unsigned)(half * (NT2 > 0 ? p.ld2b : p.ld2) + l31) * 4u;

    const int nchunks = (p.Cexp + 31) >> 5, bmask = p.nbuf - 1;
    const unsigned aoff1 = (unsigned)(half * p.ld1 + l31) * 4u, aoff2 = (unsigned)(half * p.ld2 + l31) * 4u;
    const unsigned sXstep = (unsigned)(2 * HW) * 4u;

    // first stage-A tile of this wave (tile = wave): geometry is chunk-independent
    const int q0 = wave * 32 + l31; const bool ivalid0 = q0 < PI && wave < ngi;
    const int qc0 = min(q0, PI - 1), qi0 = qc0 / (IH * p.W), qrem0 = qc0 - qi0 * (IH * p.W), ry0 = qrem0 / p.W, ix0 = qrem0 - ry0 * p.W, iy0 = iyA + ry0;
    const int ew0 = qi0 * p.HpWp + (iy0 + p.pad - ypA) * p.Wp + ix0 + p.pad;
    const unsigned xoff0 = (unsigned)((size_t)(b0 + qi0) * p.in_pitch + (size_t)iy0 * p.W + ix0 + (size_t)half * HW) * 4u;

    // depthwise convolution of k-step s of a chunk at this lane's output pixel = B operand of the project GEMM; with a second head the taps are read once and
    // convolved with both heads' weights
    auto dw = [&](const float *E, const float *Wc, const float *Wc2, int s, float &v2out) -> float {
        float w[KKP], tp[KK];
#pragma unroll
        for (int i = 0; i < KKP / 4; i++) { const float4 q4 = ((const float4 *)(Wc + 2 * s * KKP))[i]; w[4 * i] = q4.x; w[4 * i + 1] = q4.y; w[4 * i + 2] = q4.z; w[4 * i + 3] = q4.w; }
        const float *ep = E + (size_t)2 * s * p.planeT;
#pragma unroll
        for (int i = 0; i < K; i++)
#pragma unroll
            for (int j = 0; j < K; j++) tp[i * K + j] = ep[i * p.Wp + j];
        float v = w[KK];
#pragma unroll
        for (int t = 0; t < KK; t++) v = fmaf(w[t], tp[t], v);
        if (NT2 > 0) {
            float w2[KKP];
#pragma unroll
            for (int i = 0; i < KKP / 4; i++) { const float4 q4 = ((const float4 *)(Wc2 + 2 * s * KKP))[i]; w2[4 * i] = q4.x; w2[4 * i + 1] = q4.y; w2[4 * i + 2] = q4.z; w2[4 * i + 3] = q4.w; }
            float u = w2[KK];
#pragma unroll
            for (int t = 0; t < KK; t++) u = fmaf(w2[t], tp[t], u);
            v2out = sgx_irb_act(AMODE, u, p.a2c1, p.a2lo, p.a2hi, p.a2c2);
        }
        return sgx_irb_act(AMODE, v, p.a2c1, p.a2lo, p.a2hi, p.a2c2);
    };

    for (int c = -1; c < nchunks; c++) {
        const int ch1 = (c + 1) * 32, buf1 = (c + 1) & bmask;
        const bool more = c + 1 < nchunks;
        if (EXPAND && !A3 && p.w1rows && c + 2 < nchunks) stageW1(c + 2);      // into the buffer stage A read one iteration ago; visible behind this iteration's barrier
        // ---- no expand stage: the next chunk's depthwise input (first tile of this wave) is requested before this chunk's stage B and parked in registers
        float pre[EXPAND ? 1 : 16];
        if (!EXPAND && more && p.nbuf == 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) pre[r] = sgx_bld(r_in, xoff0, (unsigned)min(ch1 + 2 * r, p.Cexp - 2) / 2u * sXstep);
        }
        // ---- stage B of chunk c: depthwise into the B operand, project on the matrix cores; the depthwise value of step s + 1 is computed beside the
        // MFMAs of step s (independent instruction streams for the scheduler), the project weights are requested two steps ahead
        // With two plane buffers stage B of chunk c and stage A of chunk c + 1 touch different buffers between the same two barriers, so their order inside a
        // wave is free: waves 4..7 run A first, the others B first — at any time part of the workgroup is in the MFMA + load heavy stage A while the rest is
        // in the VALU / LDS / MFMA mix of stage B, instead of all twelve waves hitting the same phase (and its VALU-only activation tail) together.
        const int a_first = (p.nbuf == 2 && p.stagger) ? ((wave >> 2) & 1) : 0;
#pragma unroll 1
        for (int hp = 0; hp < 2; hp++) {
        if (hp == a_first && c >= 0 && owner) {
            const int ch0 = c * 32, nks = min(16, (p.Cexp - ch0) >> 1), buf = c & bmask;
            const float *E = Eb + (size_t)buf * 32 * p.planeT + e_r;
            const float *Wc = Wds + (size_t)buf * 32 * KKP + half * KKP, *Wc2 = Wds2 + (size_t)buf * 32 * KKP + half * KKP;
            const unsigned sB = (unsigned)(ch0 * p.ld2) * 4u, sstep = (unsigned)(2 * p.ld2) * 4u;
            const unsigned sBb = (unsigned)(ch0 * p.ld2b) * 4u, sstepb = (unsigned)(2 * p.ld2b) * 4u;
            // project weights: a ring of R k-steps in flight, statically indexed (the loop advances R steps per trip; Cexp is a multiple of 8: planner), so a
            // load is only waited for R steps after it was issued
#ifndef SGX_IRB_R5
#define SGX_IRB_R5 2
#endif
            constexpr int R = NT >= 5 ? SGX_IRB_R5 : 4;      // SGX_IRB_R5 / SGX_IRB_D5: A/B taps for the five-tile instantiations, whose 80 accumulator registers leave the least room (tools/ab_build.sh)
            float ar[R][NT], ar2[R][NT2A];
#pragma unroll
            for (int d = 0; d < R; d++) {
#pragma unroll
                for (int t = 0; t < NT; t++) ar[d][t] = sgx_bld(r_w2, aoff2 + 128u * t, sB + (unsigned)min(d, nks - 1) * sstep);
#pragma unroll
                for (int t = 0; t < NT2; t++) ar2[d][t] = sgx_bld(r_w2b, aoff2b + 128u * t, sBb + (unsigned)min(d, nks - 1) * sstepb);
            }
            float v2 = 0.f;
            float v = dw(E, Wc, Wc2, 0, v2);
            for (int s0 = 0; s0 < nks; s0 += R) {
#pragma unroll
                for (int d = 0; d < R; d++) {
                    const int s = s0 + d;
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[d][t], v, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NT2; t++) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar2[d][t], v2, acc2[t], 0, 0, 0);
                    const int sn_ = min(s + R, nks - 1);
#pragma unroll
                    for (int t = 0; t < NT; t++) ar[d][t] = sgx_bld(r_w2, aoff2 + 128u * t, sB + (unsigned)sn_ * sstep);
#pragma unroll
                    for (int t = 0; t < NT2; t++) ar2[d][t] = sgx_bld(r_w2b, aoff2b + 128u * t, sBb + (unsigned)sn_ * sstepb);
                    v = dw(E, Wc, Wc2, min(s + 1, nks - 1), v2);
                }
            }
        }
        if (p.nbuf == 1 && hp == 0) __syncthreads();
        // ---- stage A of chunk c + 1 into its plane buffer: expand on the matrix cores (or a plain load), activation, LDS write; the chunk's depthwise weights
        if (hp != a_first && more) {
            float *E = Eb + (size_t)buf1 * 32 * p.planeT;
            for (int i = tid; i < 8 * KKP; i += nthreads) ((float4 *)(Wds + (size_t)buf1 * 32 * KKP))[i] = ((const float4 *)(p.wdp + (size_t)ch1 * KKP))[i];
            if (NT2 > 0) for (int i = tid; i < 8 * KKP; i += nthreads) ((float4 *)(Wds2 + (size_t)buf1 * 32 * KKP))[i] = ((const float4 *)(p.wdp2 + (size_t)ch1 * KKP))[i];
            for (int tile = wave; tile < ngi; tile += nwaves) {
                const bool first = tile == wave;
                int ew; unsigned xoff; bool ivalid;
                if (first) { ew = ew0; xoff = xoff0; ivalid = ivalid0; }
                else {
                    const int q = tile * 32 + l31; ivalid = q < PI;
                    const int qc = min(q, PI - 1), qi = qc / (IH * p.W), qrem = qc - qi * (IH * p.W), ry = qrem / p.W, ix = qrem - ry * p.W, iy = iyA + ry;
                    ew = qi * p.HpWp + (iy + p.pad - ypA) * p.Wp + ix + p.pad;
                    xoff = (unsigned)((size_t)(b0 + qi) * p.in_pitch + (size_t)iy * p.W + ix + (size_t)half * HW) * 4u;
                }
                float *Ew = E + ew;
                if (EXPAND && A3) {
                    // gemm mode 2: ONLY the expand GEMM on the bf16 matrix pipes (bf16x3, sgx_det_bf16.h): its 56 fp32 MFMAs per tile (3.6 k cycles of blocked vector issue) become
                    // 42 bf16 MFMAs that run beside other waves' vector work, for 36 split instructions per k16 step; the depthwise / project / squeeze-excite stages stay as they are
                    sgx_f32x16 e;
#pragma unroll
                    for (int r = 0; r < 16; r++) e[r] = sgx_bld(r_b1, (unsigned)half * 16u, (unsigned)(ch1 + (r & 3) + 8 * (r >> 2)) * 4u);
                    const int nks1 = (p.Cin + 15) >> 4;
                    const unsigned sXrow = (unsigned)HW * 4u;
                    const unsigned xh8 = xoff - (unsigned)half * sXrow + (unsigned)(8 * half) * sXrow;          // xoff carries + half rows (k2 layout): here the half-wave starts 8 rows down
                    const unsigned xpix = xoff - (unsigned)half * sXrow;
                    auto loadX = [&](int s_, float (&dst)[8]) {
                        if (16 * s_ + 16 <= p.Cin) {
#pragma unroll
                            for (int j = 0; j < 8; j++) dst[j] = sgx_bld(r_in, xh8, (unsigned)(16 * s_ + j) * sXrow);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; j++) dst[j] = sgx_bld(r_in, xpix + (unsigned)min(16 * s_ + 8 * half + j, p.Cin - 1) * sXrow, 0u);
                        }
                    };
                    const sgx_u32x4 *w1l = (const sgx_u32x4 *)p.w1S + (size_t)half * p.ld1 + l31 + ch1;
                    auto loadW = [&](int s_, sgx_u32x4 (&dst)[3]) {
#pragma unroll
                        for (int q = 0; q < 3; q++) dst[q] = w1l[(size_t)(6 * s_ + 2 * q) * p.ld1];
                    };
                    if (p.inS) {                                           // uniform: pre-split operands, two k16 steps in flight
                        const int qq = min(tile * 32 + l31, PI - 1), qi_ = qq / (IH * p.W), qr_ = qq - qi_ * (IH * p.W), ry_ = qr_ / p.W, ix_ = qr_ - ry_ * p.W;
                        const sgx_u32x4 *xl = (const sgx_u32x4 *)p.inS + (size_t)(b0 + qi_) * (size_t)(nks1 * 6 * p.ldS) + (size_t)half * p.ldS + (size_t)((iyA + ry_) * p.W + ix_);
                        auto loadXS = [&](int s_, SgxB3 &dst) { const sgx_u32x4 *xs_ = xl + (size_t)(6 * s_) * p.ldS; dst.t0 = xs_[0]; dst.t1 = xs_[(size_t)2 * p.ldS]; dst.t2 = xs_[(size_t)4 * p.ldS]; };
                        SgxB3 xb[3]; sgx_u32x4 wr[3][3];
                        loadXS(0, xb[0]); loadW(0, wr[0]); loadXS(min(1, nks1 - 1), xb[1]); loadW(min(1, nks1 - 1), wr[1]);
                        for (int s0 = 0; s0 < nks1; s0 += 3) {
#pragma unroll
                            for (int d = 0; d < 3; d++) {
                                const int s_ = s0 + d;
                                if (s_ < nks1) {
                                    loadXS(min(s_ + 2, nks1 - 1), xb[(d + 2) % 3]); loadW(min(s_ + 2, nks1 - 1), wr[(d + 2) % 3]);
                                    e = sgx_mfma_bf16x3(wr[d][0], wr[d][1], wr[d][2], xb[d], e);
                                }
                            }
                        }
                    } else {
                    float xr[2][8]; sgx_u32x4 wr[2][3];
                    loadX(0, xr[0]); loadW(0, wr[0]);
                    for (int s0 = 0; s0 < nks1; s0 += 2) {
#pragma unroll
                        for (int d = 0; d < 2; d++) {
                            const int s_ = s0 + d;
                            if (s_ < nks1) {
                                loadX(min(s_ + 1, nks1 - 1), xr[d ^ 1]);
                                loadW(min(s_ + 1, nks1 - 1), wr[d ^ 1]);
                                const SgxB3 b = sgx_split3x8(xr[d]);
                                e = sgx_mfma_bf16x3(wr[d][0], wr[d][1], wr[d][2], b, e);
                            }
                        }
                    }
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) e[r] = sgx_irb_act(AMODE, e[r], p.a1c1, p.a1lo, p.a1hi, p.a1c2);
                    if (ivalid) {
#pragma unroll
                        for (int r = 0; r < 16; r++) Ew[(size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.planeT] = e[r];
                    }
                } else if (EXPAND) {
                    sgx_f32x16 e;
#pragma unroll
                    for (int r = 0; r < 16; r++) e[r] = sgx_bld(r_b1, (unsigned)half * 16u, (unsigned)(ch1 + (r & 3) + 8 * (r >> 2)) * 4u);
                    // operand ring: D k-steps in flight.  The k loop runs over whole groups of D steps: steps past Cin / 2 meet zero weight rows (the host pads the
                    // transposed weights to a multiple of 32 rows) and a clamped, finite input row — the per-layer kernel k_conv_pw2 pads the same way
#ifndef SGX_IRB_D5
#define SGX_IRB_D5 8
#endif
                    constexpr int D = NT >= 5 ? SGX_IRB_D5 : 8;
                    const int nks = p.Cin >> 1, nkp = (nks + D - 1) & ~(D - 1);
                    const unsigned sA = (unsigned)ch1 * 4u, sAstep = (unsigned)(2 * p.ld1) * 4u;
                    float ar[D], br[D];
                    if (p.w1rows) {                                       // uniform: weights from the chunk's LDS slice (row 2 s + half, column = the lane's channel)
                        const float *W1c = W1s + (size_t)((c + 1) & 1) * p.w1rows * 32 + half * 32 + l31;
#pragma unroll
                        for (int d = 0; d < D; d++) { ar[d] = W1c[64 * d]; br[d] = sgx_bld(r_in, xoff, (unsigned)min(d, nks - 1) * sXstep); }
                        for (int s0 = 0; s0 < nkp; s0 += D) {
#pragma unroll
                            for (int d = 0; d < D; d++) {
                                e = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[d], br[d], e, 0, 0, 0);
                                const int sn = min(s0 + d + D, nkp - 1);
                                ar[d] = W1c[64 * sn]; br[d] = sgx_bld(r_in, xoff, (unsigned)min(sn, nks - 1) * sXstep);
                            }
                        }
                    } else {
#pragma unroll
                    for (int d = 0; d < D; d++) { ar[d] = sgx_bld(r_w1, aoff1, sA + d * sAstep); br[d] = sgx_bld(r_in, xoff, (unsigned)min(d, nks - 1) * sXstep); }
                    for (int s0 = 0; s0 < nkp; s0 += D) {
#pragma unroll
                        for (int d = 0; d < D; d++) {
                            e = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[d], br[d], e, 0, 0, 0);       // the MFMA reads its operands at issue: the slot is reloaded right behind it
                            const int sn = min(s0 + d + D, nkp - 1);
                            ar[d] = sgx_bld(r_w1, aoff1, sA + (unsigned)sn * sAstep); br[d] = sgx_bld(r_in, xoff, (unsigned)min(sn, nks - 1) * sXstep);
                        }
                    }
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) e[r] = sgx_irb_act(AMODE, e[r], p.a1c1, p.a1lo, p.a1hi, p.a1c2);
                    if (ivalid) {                                         // rows past Cexp of the last chunk are written but never read (stage B stops at Cexp)
#pragma unroll
                        for (int r = 0; r < 16; r++) Ew[(size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.planeT] = e[r];
                    }
                } else {
                    // depthwise input from global memory: lane (half, pixel) loads channels ch1 + 2 r + half, r = 0..15 (two coalesced rows per load)
                    float v[16];
                    if (first && p.nbuf == 2) {
#pragma unroll
                        for (int r = 0; r < 16; r++) v[r] = pre[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r++) v[r] = sgx_bld(r_in, xoff, (unsigned)min(ch1 + 2 * r, p.Cexp - 2) / 2u * sXstep);
                    }
                    if (ivalid) {
#pragma unroll
                        for (int r = 0; r < 16; r++) Ew[(size_t)(2 * r + half) * p.planeT] = v[r];
                    }
                }
            }
        }
        }
        __syncthreads();
    }

    // ---- epilogue: [squeeze-excite gate] [+ residual], store
    if (!owner) return;
    if (NQ > 0) {
        // Both 1x1 convolutions of the gate run out of registers in groups of four k-steps (one register quad of a tile = 8 rows); the A operands of the
        // next group are in flight while the current group's MFMAs issue, and a scheduling barrier per group keeps the compiler from hoisting every load
        // of the unrolled chain to the top (hundreds of live registers).
        const sgx_rsrc r_q1 = sgx_mkrsrc(p.wq1T), r_q2 = sgx_mkrsrc(p.wq2T), r_bq1 = sgx_mkrsrc_n(p.bq1, p.Cq * 4), r_bq2 = sgx_mkrsrc_n(p.bq2, p.Cout * 4);
        constexpr int NQ1 = NQ > 0 ? NQ : 1;
        sgx_f32x16 qa[NQ1];
#pragma unroll
        for (int u = 0; u < NQ; u++)
#pragma unroll
            for (int r = 0; r < 16; r++) qa[u][r] = SGX_IRB_BIAS(r_bq1, u, r, half);
        const unsigned aoffq1 = (unsigned)(half * p.ldq1 + l31) * 4u, aoffq2 = (unsigned)(half * p.ldq2 + l31) * 4u;
        const unsigned sq1 = (unsigned)(2 * p.ldq1) * 4u, sq2 = (unsigned)(2 * p.ldq2) * 4u;          // byte step of one k-step (two weight rows)
        {
            float an[4][NQ1];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int u = 0; u < NQ; u++) an[j][u] = sgx_bld(r_q1, aoffq1 + 128u * u, (unsigned)j * sq1);
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int g = 4 * t + m;                              // k-steps 4 g .. 4 g + 3 = rows 8 g .. 8 g + 7 (Cout is a multiple of 8: planner)
                    if (8 * g < p.Cout) {
                        float cur[4][NQ1];
#pragma unroll
                        for (int j = 0; j < 4; j++)
#pragma unroll
                            for (int u = 0; u < NQ; u++) cur[j][u] = an[j][u];
                        if (8 * (g + 1) < p.Cout) {
#pragma unroll
                            for (int j = 0; j < 4; j++)
#pragma unroll
                                for (int u = 0; u < NQ; u++) an[j][u] = sgx_bld(r_q1, aoffq1 + 128u * u, (unsigned)(4 * (g + 1) + j) * sq1);
                        }
                        const auto r01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[t][4 * m]), __float_as_uint(acc[t][4 * m + 1]), false, false);
                        const auto r23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[t][4 * m + 2]), __float_as_uint(acc[t][4 * m + 3]), false, false);
                        const float bv[4] = { __uint_as_float(r01[0]), __uint_as_float(r23[0]), __uint_as_float(r01[1]), __uint_as_float(r23[1]) };
#pragma unroll
                        for (int j = 0; j < 4; j++)
#pragma unroll
                            for (int u = 0; u < NQ; u++) qa[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[j][u], bv[j], qa[u], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        float qb[NQ1][16];
#pragma unroll
        for (int u = 0; u < NQ; u++) {
#pragma unroll
            for (int r = 0; r < 16; r++) qa[u][r] = sgx_clipf(qa[u][r], p.qlo, p.qhi);
            sgx_irb_d2b(qa[u], qb[u]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            sgx_f32x16 ga;
#pragma unroll
            for (int r = 0; r < 16; r++) ga[r] = SGX_IRB_BIAS(r_bq2, t, r, half);
            float an[4];
#pragma unroll
            for (int j = 0; j < 4; j++) an[j] = sgx_bld(r_q2, aoffq2 + 128u * t, (unsigned)j * sq2);
#pragma unroll
            for (int g = 0; g < 4 * NQ; g++) {
                if (8 * g < p.Cq) {
                    float cur[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) cur[j] = an[j];
                    if (8 * (g + 1) < p.Cq) {
#pragma unroll
                        for (int j = 0; j < 4; j++) an[j] = sgx_bld(r_q2, aoffq2 + 128u * t, (unsigned)(4 * (g + 1) + j) * sq2);
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (2 * (4 * g + j) < p.Cq) ga = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[j], qb[g >> 2][4 * (g & 3) + j], ga, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { float u_ = ga[r] + p.gc1; u_ = sgx_clipf(u_, p.glo, p.ghi); u_ = sgx_div_c2(u_, p.gc2); acc[t][r] = u_ * acc[t][r]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // store (+ residual): buffer addressing = lane offset (image, pixel, half-wave row offset) + per-register scalar / immediate row offset
    if (ovalid) {
        const sgx_rsrc r_out = sgx_mkrsrc(p.out), r_res = sgx_mkrsrc(p.has_res ? (const void *)p.res : (const void *)p.out);
        const unsigned rvo = (unsigned)((size_t)(b0 + og_img) * p.res_pitch + opix + (size_t)4 * half * HWo) * 4u;
        const unsigned ovo = p.hwc ? (unsigned)((size_t)(b0 + og_img) * p.out_pitch + (size_t)p.hwc_off + (size_t)opix * p.Cout + 4 * half) * 4u
                                   : (unsigned)((size_t)(b0 + og_img) * p.out_pitch + opix + (size_t)4 * half * HWo) * 4u;
        const unsigned rowstep = p.hwc ? 4u : (unsigned)HWo * 4u;
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rb = 32 * t + (r & 3) + 8 * (r >> 2);
                if (rb + 4 * half < p.Cout) {
                    float v = acc[t][r];
                    if (p.has_res) v = v + sgx_bld(r_res, rvo, (unsigned)rb * (unsigned)HWo * 4u);
                    sgx_bst(r_out, ovo, (unsigned)rb * rowstep, v);
                }
            }
        }
    }
    if (NT2 > 0 && ovalid) {                          // second head: HWC store into its own concat buffer
        const sgx_rsrc r_o2 = sgx_mkrsrc(p.out2);
        const unsigned ovo2 = (unsigned)((size_t)(b0 + og_img) * p.out2_pitch + (size_t)p.hwc_off2 + (size_t)opix * p.Cout2 + 4 * half) * 4u;
#pragma unroll
        for (int t = 0; t < NT2; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rb = 32 * t + (r & 3) + 8 * (r >> 2);
                if (rb + 4 * half < p.Cout2) sgx_bst(r_o2, ovo2, (unsigned)rb * 4u, acc2[t][r]);
            }
        }
    }
}
#endif

#if !defined(SGX_EMU) && defined(SGX_DEBUG_TAPS)      /* k_irb3: built, correct, slower than k_irb (profiles/HISTORY_r1-r4.md): tap build only (SGX_DET_IRB3=1) */
// ---------------------------------------------------------------------------------------------
// k_irb3: the block of k_irb with every matrix product on v_mfma_f32_32x32x16_bf16 (bf16x3, sgx_det_bf16.h).  Same work decomposition, LDS planes, barriers, depthwise
// arithmetic (fmaf in tap order) and epilogue; what changes is the operand shape of the three GEMMs:
//   expand   a k16 step = the lane's eight input channels 16 s + 8 half + j of its input pixel (eight row loads), split once, against three dwordx4 weight loads
//   project  lane (half h, pixel) computes the depthwise outputs of channels 16 s + 8 h + j, j = 0..7, at ITS output pixel (was: one channel per k2 step): after the split
//            they are the B operand of one k16 step for all NT output tiles
//   squeeze-excite  accumulator rows -> B layout with FOUR v_permlane32_swap per k16 step (registers 8 q + i <-> 8 q + 4 + i: the lower half-wave ends up with rows
//            16 q + 0..7, the upper with 16 q + 8..15), then the split
// NQS = k16 steps of the squeeze width Cq (the launch checks it).
// ---------------------------------------------------------------------------------------------
// Activation of the bf16x3 kernels: the same formula with the division by c2 as q0 = u rc, q = fma(fma(-q0, c2, u), rc, q0) (rc = RN(1 / c2)) WITHOUT the guard + IEEE
// fall-back of sgx_div_c2: that guard is a wave-uniform branch per activation, which cuts the depthwise stage into one basic block per channel (no overlap of one
// channel's LDS reads with another's arithmetic).  The unguarded form equals u / c2 unless |u| < 2^-123 (then it is off by a few 2^-149) — irrelevant in a plan that makes
// no bit-exactness claim; the exact-fp32 plan keeps sgx_div_c2.
SGX_DEV float sgx_irb_act_fast(int mode, float v, float c1, float lo, float hi, float c2, float rc)
{
    if (mode == SGX_EMODE_HSWISH) { float u = v + c1; u = sgx_clipf(u, lo, hi); u = u * v; const float q0 = u * rc; return fmaf(fmaf(-q0, c2, u), rc, q0); }
    return sgx_clipf(v, lo, hi);
}
SGX_DEV void sgx_irb_d2b16(const sgx_f32x16 &d, int q, float (&b)[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[8 * q + i]), __float_as_uint(d[8 * q + 4 + i]), false, false);
        b[i] = __uint_as_float(sw[0]); b[4 + i] = __uint_as_float(sw[1]);
    }
}

template <int K, int S, int NT, int NQ, bool EXPAND, bool HS, int NT2>
__global__ void __launch_bounds__(768) k_irb3(SgxIrb p)
{
    extern __shared__ __attribute__((aligned(16))) float sgx_irb_smem[];
    constexpr int KK = K * K, KKP = SGX_IRB_KKP(K);
    constexpr int AMODE = HS ? SGX_EMODE_HSWISH : SGX_EMODE_ACT;
    constexpr int NQS = NQ == 0 ? 1 : (NQ == 2 ? 3 : (NT == 2 ? 1 : 2));
    float *Eb = sgx_irb_smem;                                         // [nbuf][32][planeT]
    float *Wds = sgx_irb_smem + (size_t)p.nbuf * 32 * p.planeT;       // [nbuf][32][KKP]
    float *Wds2 = Wds + (size_t)p.nbuf * 32 * KKP;                    // second head's depthwise weights (NT2 > 0)
    const int tid = (int)threadIdx.x, nthreads = (int)blockDim.x, wave = tid >> 6, nwaves = nthreads >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W, HWo = p.Ho * p.Wo;
    int b0, nimg, oy0, OH;
    if (p.nbands > 1) { b0 = (int)blockIdx.x / p.nbands; nimg = 1; oy0 = ((int)blockIdx.x - b0 * p.nbands) * p.OH; OH = min(p.OH, p.Ho - oy0); }
    else { b0 = (int)blockIdx.x * p.G; nimg = min(p.G, p.batch - b0); oy0 = 0; OH = p.Ho; }
    const int ypA = oy0 * S;
    const int iyA = max(0, ypA - p.pad), iyB = min(p.H, (oy0 + OH - 1) * S + K - p.pad), IH = iyB - iyA;
    const int PI = nimg * IH * p.W, PO = nimg * OH * p.Wo, ngi = (PI + 31) >> 5, ngo = (PO + 31) >> 5;
    const bool owner = wave < ngo;
    const int og = wave * 32 + l31; const bool ovalid = og < PO;
    const int oc_ = min(og, PO - 1), og_img = oc_ / (OH * p.Wo), orem = oc_ - og_img * (OH * p.Wo), oyl = orem / p.Wo, ox = orem - oyl * p.Wo;
    const int e_r = 8 * half * p.planeT + og_img * p.HpWp + oyl * S * p.Wp + ox * S;            // depthwise read base of channel 8 half (channel 16 s + j: + (16 s + j) planeT)
    const unsigned opix = (unsigned)((oy0 + oyl) * p.Wo + ox);
    const sgx_rsrc r_in = sgx_mkrsrc(p.in), r_b1 = sgx_mkrsrc_n(EXPAND ? p.b1 : p.b2, (EXPAND ? p.Cexp : p.Cout) * 4);

    {
        float4 *z = (float4 *)Eb; const int nz = p.nbuf * 8 * p.planeT;
        for (int i = tid; i < nz; i += nthreads) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();                                                  // the zero-fill before any wave writes a tile into the planes
    sgx_f32x16 acc[NT];
    {
        const sgx_rsrc r_b2 = sgx_mkrsrc_n(p.b2, p.Cout * 4);
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = SGX_IRB_BIAS(r_b2, t, r, half);
    }
    constexpr int NT2A = NT2 > 0 ? NT2 : 1;
    sgx_f32x16 acc2[NT2A];
    if (NT2 > 0) {
        const sgx_rsrc r_b2b = sgx_mkrsrc_n(p.b2b, p.Cout2 * 4);
#pragma unroll
        for (int t = 0; t < NT2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[t][r] = SGX_IRB_BIAS(r_b2b, t, r, half);
    }
    const int nchunks = (p.Cexp + 31) >> 5, bmask = p.nbuf - 1;
    // split-weight operand pointers of this lane: + (6 ks + 2 term) ld + 32 tile   (in 16-byte units)
    const sgx_u32x4 *a1p = (const sgx_u32x4 *)(EXPAND ? p.w1S : p.w2S) + (size_t)half * (EXPAND ? p.ld1 : p.ld2) + l31;
    const sgx_u32x4 *a2p = (const sgx_u32x4 *)p.w2S + (size_t)half * p.ld2 + l31;
    const sgx_u32x4 *a2bp = (const sgx_u32x4 *)(NT2 > 0 ? p.w2Sb : p.w2S) + (size_t)half * (NT2 > 0 ? p.ld2b : p.ld2) + l31;
    const unsigned sXrow = (unsigned)HW * 4u;
    const float rc1 = 1.0f / p.a1c2, rc2 = 1.0f / p.a2c2, rcg = 1.0f / p.gc2;       // wave-uniform reciprocals of the activation divisors (sgx_irb_act_fast)

    // first stage-A tile of this wave (tile = wave): geometry is chunk-independent
    const int q0 = wave * 32 + l31; const bool ivalid0 = q0 < PI && wave < ngi;
    const int qc0 = min(q0, PI - 1), qi0 = qc0 / (IH * p.W), qrem0 = qc0 - qi0 * (IH * p.W), ry0 = qrem0 / p.W, ix0 = qrem0 - ry0 * p.W, iy0 = iyA + ry0;
    const int ew0 = qi0 * p.HpWp + (iy0 + p.pad - ypA) * p.Wp + ix0 + p.pad;
    const unsigned xpix0 = (unsigned)((size_t)(b0 + qi0) * p.in_pitch + (size_t)iy0 * p.W + ix0) * 4u;               // channel 0 of the lane's input pixel
    const unsigned xoff0 = xpix0 + (unsigned)half * sXrow;                                                            // no-expand staging: channel pair layout of k_irb

    // depthwise convolution of the lane's eight channels of k16 step s at its output pixel, split into the three bf16 operands pair by pair (a channel pair is
    // one register of each term); a scheduling fence per channel (5 x 5) or pair (3 x 3) keeps the live set at one or two channels' taps and weights
    auto dw8 = [&](const float *E, const float *Wc, const float *Wc2, int s, SgxB3 &bo, SgxB3 &bo2) {
#pragma unroll
        for (int jp = 0; jp < 4; jp++) {
            float vv[2], vv2[2];
#pragma unroll
            for (int jh = 0; jh < 2; jh++) {
                const int j = 2 * jp + jh;
                float w[KKP], tp[KK];
                const float *wj = Wc + (16 * s + j) * KKP;
#pragma unroll
                for (int i = 0; i < KKP / 4; i++) { const float4 q4 = ((const float4 *)wj)[i]; w[4 * i] = q4.x; w[4 * i + 1] = q4.y; w[4 * i + 2] = q4.z; w[4 * i + 3] = q4.w; }
                const float *ep = E + (size_t)(16 * s + j) * p.planeT;
#pragma unroll
                for (int i = 0; i < K; i++)
#pragma unroll
                    for (int jj = 0; jj < K; jj++) tp[i * K + jj] = ep[i * p.Wp + jj];
                float v = w[KK];
#pragma unroll
                for (int t = 0; t < KK; t++) v = fmaf(w[t], tp[t], v);
                vv[jh] = sgx_irb_act_fast(AMODE, v, p.a2c1, p.a2lo, p.a2hi, p.a2c2, rc2);
                if (NT2 > 0) {
                    float w2[KKP];
                    const float *wj2 = Wc2 + (16 * s + j) * KKP;
#pragma unroll
                    for (int i = 0; i < KKP / 4; i++) { const float4 q4 = ((const float4 *)wj2)[i]; w2[4 * i] = q4.x; w2[4 * i + 1] = q4.y; w2[4 * i + 2] = q4.z; w2[4 * i + 3] = q4.w; }
                    float u = w2[KK];
#pragma unroll
                    for (int t = 0; t < KK; t++) u = fmaf(w2[t], tp[t], u);
                    vv2[jh] = sgx_irb_act_fast(AMODE, u, p.a2c1, p.a2lo, p.a2hi, p.a2c2, rc2);
                }
            }
            { unsigned a0, a1, a2; sgx_split3(vv[0], vv[1], a0, a1, a2); bo.t0[jp] = a0; bo.t1[jp] = a1; bo.t2[jp] = a2; }
            if (NT2 > 0) { unsigned a0, a1, a2; sgx_split3(vv2[0], vv2[1], a0, a1, a2); bo2.t0[jp] = a0; bo2.t1[jp] = a1; bo2.t2[jp] = a2; }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (int c = -1; c < nchunks; c++) {
        const int ch1 = (c + 1) * 32, buf1 = (c + 1) & bmask;
        const bool more = c + 1 < nchunks;
        float pre[EXPAND ? 1 : 16];
        if (!EXPAND && more && p.nbuf == 2) {
#pragma unroll
            for (int r = 0; r < 16; r++) pre[r] = sgx_bld(r_in, xoff0, (unsigned)min(ch1 + 2 * r, p.Cexp - 2) / 2u * (2u * sXrow));
        }
        const int a_first = (p.nbuf == 2 && p.stagger) ? ((wave >> 2) & 1) : 0;
#pragma unroll 1
        for (int hp = 0; hp < 2; hp++) {
        if (hp == a_first && c >= 0 && owner && !(p.dbg & 16)) {
            // ---- stage B of chunk c: depthwise -> split -> project on the bf16 matrix pipes
            const int ch0 = c * 32, nks = min(2, (p.Cexp - ch0 + 15) >> 4), buf = c & bmask;
            const float *E = Eb + (size_t)buf * 32 * p.planeT + e_r;
            const float *Wc = Wds + (size_t)buf * 32 * KKP + 8 * half * KKP, *Wc2 = Wds2 + (size_t)buf * 32 * KKP + 8 * half * KKP;
            for (int s = 0; s < nks; s++) {
                const sgx_u32x4 *wa = a2p + (size_t)(6 * (2 * c + s)) * p.ld2, *wb = a2bp + (size_t)(6 * (2 * c + s)) * (NT2 > 0 ? p.ld2b : p.ld2);
                // A operands: tile t + 1 is requested while tile t multiplies (two register sets; one for the five-tile blocks, whose accumulators leave no room)
                constexpr int AR = NT >= 5 ? 1 : 2;
                sgx_u32x4 ar[AR][3];
#pragma unroll
                for (int q = 0; q < 3; q++) ar[0][q] = wa[(size_t)(2 * q) * p.ld2];
                SgxB3 b, bb;
                if (p.dbg & 1) { b.t0 = ar[0][0]; b.t1 = ar[0][1]; b.t2 = ar[0][2]; bb = b; } else
                dw8(E, Wc, Wc2, s, b, bb);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    if (AR == 2) {
                        if (t + 1 < NT) {
#pragma unroll
                            for (int q = 0; q < 3; q++) ar[(t + 1) % AR][q] = wa[(size_t)(2 * q) * p.ld2 + 32 * (t + 1)];
                        } else if (NT2 > 0) {
#pragma unroll
                            for (int q = 0; q < 3; q++) ar[(t + 1) % AR][q] = wb[(size_t)(2 * q) * p.ld2b];
                        }
                    } else if (t > 0) {
#pragma unroll
                        for (int q = 0; q < 3; q++) ar[0][q] = wa[(size_t)(2 * q) * p.ld2 + 32 * t];
                    }
                    if (p.dbg & 2) { acc[t][0] += __uint_as_float(ar[t % AR][0][0] ^ b.t0[0] ^ b.t1[1] ^ b.t2[2]); } else
                    acc[t] = sgx_mfma_bf16x3(ar[t % AR][0], ar[t % AR][1], ar[t % AR][2], b, acc[t]);
                }
                if (NT2 > 0) {
#pragma unroll
                    for (int t = 0; t < NT2; t++) {
                        if (AR == 1 || t > 0) {
#pragma unroll
                            for (int q = 0; q < 3; q++) ar[(NT + t) % AR][q] = wb[(size_t)(2 * q) * p.ld2b + 32 * t];
                        }
                        acc2[t] = sgx_mfma_bf16x3(ar[(NT + t) % AR][0], ar[(NT + t) % AR][1], ar[(NT + t) % AR][2], bb, acc2[t]);
                    }
                }
            }
        }
        if (p.nbuf == 1 && hp == 0) __syncthreads();
        // ---- stage A of chunk c + 1 into its plane buffer
        if (hp != a_first && more) {
            float *E = Eb + (size_t)buf1 * 32 * p.planeT;
            for (int i = tid; i < 8 * KKP; i += nthreads) ((float4 *)(Wds + (size_t)buf1 * 32 * KKP))[i] = ((const float4 *)(p.wdp + (size_t)ch1 * KKP))[i];
            if (NT2 > 0) for (int i = tid; i < 8 * KKP; i += nthreads) ((float4 *)(Wds2 + (size_t)buf1 * 32 * KKP))[i] = ((const float4 *)(p.wdp2 + (size_t)ch1 * KKP))[i];
            for (int tile = wave; tile < ((p.dbg & 32) ? 0 : ngi); tile += nwaves) {
                const bool first = tile == wave;
                int ew; unsigned xpix; bool ivalid;
                if (first) { ew = ew0; xpix = xpix0; ivalid = ivalid0; }
                else {
                    const int q = tile * 32 + l31; ivalid = q < PI;
                    const int qc = min(q, PI - 1), qi = qc / (IH * p.W), qrem = qc - qi * (IH * p.W), ry = qrem / p.W, ix = qrem - ry * p.W, iy = iyA + ry;
                    ew = qi * p.HpWp + (iy + p.pad - ypA) * p.Wp + ix + p.pad;
                    xpix = (unsigned)((size_t)(b0 + qi) * p.in_pitch + (size_t)iy * p.W + ix) * 4u;
                }
                float *Ew = E + ew;
                if (EXPAND && (p.dbg & 64)) {
                    // expand stage on the fp32 matrix cores exactly as k_irb (operand ring of 8 k2 steps): the variant "bf16x3 for the project / squeeze-excite GEMMs only"
                    sgx_f32x16 e;
#pragma unroll
                    for (int r = 0; r < 16; r++) e[r] = sgx_bld(r_b1, (unsigned)half * 16u, (unsigned)(ch1 + (r & 3) + 8 * (r >> 2)) * 4u);
                    const sgx_rsrc r_w1 = sgx_mkrsrc(p.w1T);
                    const unsigned aoff1 = (unsigned)(half * p.ld1 + l31) * 4u, xoff = xpix + (unsigned)half * sXrow, sXstep = 2u * sXrow;
                    constexpr int D = 8;
                    const int nks = p.Cin >> 1, nkp = (nks + D - 1) & ~(D - 1);
                    const unsigned sA = (unsigned)ch1 * 4u, sAstep = (unsigned)(2 * p.ld1) * 4u;
                    float ar[D], br[D];
#pragma unroll
                    for (int d = 0; d < D; d++) { ar[d] = sgx_bld(r_w1, aoff1, sA + d * sAstep); br[d] = sgx_bld(r_in, xoff, (unsigned)min(d, nks - 1) * sXstep); }
                    for (int s0 = 0; s0 < nkp; s0 += D) {
#pragma unroll
                        for (int d = 0; d < D; d++) {
                            e = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[d], br[d], e, 0, 0, 0);
                            const int sn = min(s0 + d + D, nkp - 1);
                            ar[d] = sgx_bld(r_w1, aoff1, sA + (unsigned)sn * sAstep); br[d] = sgx_bld(r_in, xoff, (unsigned)min(sn, nks - 1) * sXstep);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) e[r] = sgx_irb_act_fast(AMODE, e[r], p.a1c1, p.a1lo, p.a1hi, p.a1c2, rc1);
                    if (ivalid) {
#pragma unroll
                        for (int r = 0; r < 16; r++) Ew[(size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.planeT] = e[r];
                    }
                } else if (EXPAND) {
                    sgx_f32x16 e;
#pragma unroll
                    for (int r = 0; r < 16; r++) e[r] = sgx_bld(r_b1, (unsigned)half * 16u, (unsigned)(ch1 + (r & 3) + 8 * (r >> 2)) * 4u);
                    const int nks1 = (p.Cin + 15) >> 4;
                    const unsigned xh8 = xpix + (unsigned)(8 * half) * sXrow;
                    // input channels 16 s + 8 half + j of the lane's input pixel; the last step of a block whose Cin is not a multiple of 16 clamps its rows per lane
                    // (they meet zero weight rows)
                    auto loadX = [&](int s, float (&dst)[8]) {
                        if (16 * s + 16 <= p.Cin) {
#pragma unroll
                            for (int j = 0; j < 8; j++) dst[j] = sgx_bld(r_in, xh8, (unsigned)(16 * s + j) * sXrow);
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; j++) dst[j] = sgx_bld(r_in, xpix + (unsigned)min(16 * s + 8 * half + j, p.Cin - 1) * sXrow, 0u);
                        }
                    };
                    const sgx_u32x4 *w1l = a1p + ch1;
                    auto loadW = [&](int s, sgx_u32x4 (&dst)[3]) {
#pragma unroll
                        for (int q = 0; q < 3; q++) dst[q] = w1l[(size_t)(6 * s + 2 * q) * p.ld1];
                    };
                    // input rows and weights one k16 step ahead (two register sets each)
                    float xr[2][8]; sgx_u32x4 wr[2][3];
                    loadX(0, xr[0]); loadW(0, wr[0]);
                    for (int s0 = 0; s0 < nks1; s0 += 2) {
#pragma unroll
                        for (int d = 0; d < 2; d++) {
                            const int s = s0 + d;
                            if (s < nks1) {
                                loadX(min(s + 1, nks1 - 1), xr[d ^ 1]);
                                loadW(min(s + 1, nks1 - 1), wr[d ^ 1]);
                                SgxB3 b;
                                if (p.dbg & 8) { b.t0[0] = __float_as_uint(xr[d][0]); b.t0[1] = __float_as_uint(xr[d][1]); b.t0[2] = __float_as_uint(xr[d][2]); b.t0[3] = __float_as_uint(xr[d][3]); b.t1[0] = __float_as_uint(xr[d][4]); b.t1[1] = __float_as_uint(xr[d][5]); b.t1[2] = __float_as_uint(xr[d][6]); b.t1[3] = __float_as_uint(xr[d][7]); b.t2 = b.t0; }
                                else b = sgx_split3x8(xr[d]);
                                if (p.dbg & 4) e[0] += __uint_as_float(wr[d][0][0] ^ wr[d][1][1] ^ wr[d][2][2] ^ b.t0[0] ^ b.t1[1] ^ b.t2[2]);
                                else e = sgx_mfma_bf16x3(wr[d][0], wr[d][1], wr[d][2], b, e);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) e[r] = sgx_irb_act_fast(AMODE, e[r], p.a1c1, p.a1lo, p.a1hi, p.a1c2, rc1);
                    if (ivalid) {
#pragma unroll
                        for (int r = 0; r < 16; r++) Ew[(size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * p.planeT] = e[r];
                    }
                } else {
                    const unsigned xoff = xpix + (unsigned)half * sXrow;
                    float v[16];
                    if (first && p.nbuf == 2) {
#pragma unroll
                        for (int r = 0; r < 16; r++) v[r] = pre[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; r++) v[r] = sgx_bld(r_in, xoff, (unsigned)min(ch1 + 2 * r, p.Cexp - 2) / 2u * (2u * sXrow));
                    }
                    if (ivalid) {
#pragma unroll
                        for (int r = 0; r < 16; r++) Ew[(size_t)(2 * r + half) * p.planeT] = v[r];
                    }
                }
            }
        }
        }
        __syncthreads();
    }

    // ---- epilogue: [squeeze-excite gate] [+ residual], store
    if (!owner) return;
    if (NQ > 0) {
        const sgx_rsrc r_bq1 = sgx_mkrsrc_n(p.bq1, p.Cq * 4), r_bq2 = sgx_mkrsrc_n(p.bq2, p.Cout * 4);
        constexpr int NQ1 = NQ > 0 ? NQ : 1;
        const sgx_u32x4 *q1p = (const sgx_u32x4 *)p.wq1S + (size_t)half * p.ldq1 + l31, *q2p = (const sgx_u32x4 *)p.wq2S + (size_t)half * p.ldq2 + l31;
        sgx_f32x16 qa[NQ1];
#pragma unroll
        for (int u = 0; u < NQ; u++)
#pragma unroll
            for (int r = 0; r < 16; r++) qa[u][r] = SGX_IRB_BIAS(r_bq1, u, r, half);
        // squeeze: K = Cout, two k16 steps per accumulator tile
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int ks = 2 * t + q;
                if (16 * ks < p.Cout) {
                    sgx_u32x4 aq[NQ1][3];
#pragma unroll
                    for (int u = 0; u < NQ; u++)
#pragma unroll
                        for (int m = 0; m < 3; m++) aq[u][m] = q1p[(size_t)(6 * ks + 2 * m) * p.ldq1 + 32 * u];
                    float bv[8];
                    sgx_irb_d2b16(acc[t], q, bv);
                    const SgxB3 b = sgx_split3x8(bv);
#pragma unroll
                    for (int u = 0; u < NQ; u++) qa[u] = sgx_mfma_bf16x3(aq[u][0], aq[u][1], aq[u][2], b, qa[u]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        SgxB3 qb[NQS];
#pragma unroll
        for (int g = 0; g < NQS; g++) {
            float bv[8];
#pragma unroll
            for (int r = 0; r < 8; r++) { qa[g >> 1][8 * (g & 1) + r] = sgx_clipf(qa[g >> 1][8 * (g & 1) + r], p.qlo, p.qhi); }
            sgx_irb_d2b16(qa[g >> 1], g & 1, bv);
            qb[g] = sgx_split3x8(bv);
        }
        __builtin_amdgcn_sched_barrier(0);
        // excite: K = Cq (NQS k16 steps), gate, multiply
#pragma unroll
        for (int t = 0; t < NT; t++) {
            sgx_f32x16 ga;
#pragma unroll
            for (int r = 0; r < 16; r++) ga[r] = SGX_IRB_BIAS(r_bq2, t, r, half);
#pragma unroll
            for (int g = 0; g < NQS; g++) {
                sgx_u32x4 ae[3];
#pragma unroll
                for (int m = 0; m < 3; m++) ae[m] = q2p[(size_t)(6 * g + 2 * m) * p.ldq2 + 32 * t];
                ga = sgx_mfma_bf16x3(ae[0], ae[1], ae[2], qb[g], ga);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { float u_ = ga[r] + p.gc1; u_ = sgx_clipf(u_, p.glo, p.ghi); const float q0 = u_ * rcg; u_ = fmaf(fmaf(-q0, p.gc2, u_), rcg, q0); acc[t][r] = u_ * acc[t][r]; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (ovalid) {
        const sgx_rsrc r_out = sgx_mkrsrc(p.out), r_res = sgx_mkrsrc(p.has_res ? (const void *)p.res : (const void *)p.out);
        const unsigned rvo = (unsigned)((size_t)(b0 + og_img) * p.res_pitch + opix + (size_t)4 * half * HWo) * 4u;
        const unsigned ovo = p.hwc ? (unsigned)((size_t)(b0 + og_img) * p.out_pitch + (size_t)p.hwc_off + (size_t)opix * p.Cout + 4 * half) * 4u
                                   : (unsigned)((size_t)(b0 + og_img) * p.out_pitch + opix + (size_t)4 * half * HWo) * 4u;
        const unsigned rowstep = p.hwc ? 4u : (unsigned)HWo * 4u;
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rb = 32 * t + (r & 3) + 8 * (r >> 2);
                if (rb + 4 * half < p.Cout) {
                    float v = acc[t][r];
                    if (p.has_res) v = v + sgx_bld(r_res, rvo, (unsigned)rb * (unsigned)HWo * 4u);
                    sgx_bst(r_out, ovo, (unsigned)rb * rowstep, v);
                }
            }
        }
    }
    if (NT2 > 0 && ovalid) {
        const sgx_rsrc r_o2 = sgx_mkrsrc(p.out2);
        const unsigned ovo2 = (unsigned)((size_t)(b0 + og_img) * p.out2_pitch + (size_t)p.hwc_off2 + (size_t)opix * p.Cout2 + 4 * half) * 4u;
#pragma unroll
        for (int t = 0; t < NT2; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rb = 32 * t + (r & 3) + 8 * (r >> 2);
                if (rb + 4 * half < p.Cout2) sgx_bst(r_o2, ovo2, (unsigned)rb * 4u, acc2[t][r]);
            }
        }
    }
}
#endif

#ifdef SGX_EMU
// kernel-logic emulator: the block as scalar fmaf chains in the device kernel's order (bias first, k ascending, taps (i, j) ascending; taps in the zero
// padding add an exact zero), one call per image.
static void sgx_irb_emu(const SgxIrb &p, int b)
{
    const int HW = p.H * p.W, HWo = p.Ho * p.Wo, KK = p.K * p.K;
    const float *X = p.in + (size_t)b * p.in_pitch;
    std::vector<float> E((size_t)p.Cexp * HW), Dw((size_t)p.Cexp * HWo), Y((size_t)p.Cout * HWo), Q((size_t)std::max(p.Cq, 1) * HWo);
    for (int m = 0; m < p.Cexp; m++) for (int q = 0; q < HW; q++) {
        if (!p.has_expand) { E[(size_t)m * HW + q] = X[(size_t)m * HW + q]; continue; }
        float s = p.b1[m];
        for (int k = 0; k < p.Cin; k++) s = fmaf(p.w1[(size_t)m * p.Cin + k], X[(size_t)k * HW + q], s);
        E[(size_t)m * HW + q] = sgx_irb_act(p.act1, s, p.a1c1, p.a1lo, p.a1hi, p.a1c2);
    }
    for (int m = 0; m < p.Cexp; m++) for (int oy = 0; oy < p.Ho; oy++) for (int ox = 0; ox < p.Wo; ox++) {
        float s = p.bd[m];
        for (int i = 0; i < p.K; i++) for (int j = 0; j < p.K; j++) {
            const int iy = oy * p.S - p.pad + i, ix = ox * p.S - p.pad + j;
            const float x = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? E[(size_t)m * HW + iy * p.W + ix] : 0.f;
            s = fmaf(p.wd[(size_t)m * KK + i * p.K + j], x, s);
        }
        Dw[(size_t)m * HWo + oy * p.Wo + ox] = sgx_irb_act(p.act2, s, p.a2c1, p.a2lo, p.a2hi, p.a2c2);
    }
    for (int o = 0; o < p.Cout; o++) for (int q = 0; q < HWo; q++) {
        float s = p.b2[o];
        for (int k = 0; k < p.Cexp; k++) s = fmaf(p.w2[(size_t)o * p.Cexp + k], Dw[(size_t)k * HWo + q], s);
        Y[(size_t)o * HWo + q] = s;
    }
    if (p.Cq > 0) {
        for (int u = 0; u < p.Cq; u++) for (int q = 0; q < HWo; q++) {
            float s = p.bq1[u];
            for (int k = 0; k < p.Cout; k++) s = fmaf(p.wq1[(size_t)u * p.Cout + k], Y[(size_t)k * HWo + q], s);
            Q[(size_t)u * HWo + q] = fminf(fmaxf(s, p.qlo), p.qhi);
        }
        for (int o = 0; o < p.Cout; o++) for (int q = 0; q < HWo; q++) {
            float s = p.bq2[o];
            for (int k = 0; k < p.Cq; k++) s = fmaf(p.wq2[(size_t)o * p.Cq + k], Q[(size_t)k * HWo + q], s);
            float u = s + p.gc1; u = fminf(fmaxf(u, p.glo), p.ghi); u = u / p.gc2;
            Y[(size_t)o * HWo + q] = u * Y[(size_t)o * HWo + q];
        }
    }
    for (int o = 0; o < p.Cout; o++) for (int q = 0; q < HWo; q++) {
        float v = Y[(size_t)o * HWo + q];
        if (p.has_res) v = v + p.res[(size_t)b * p.res_pitch + (size_t)o * HWo + q];
        if (p.hwc) p.out[(size_t)b * p.out_pitch + (size_t)p.hwc_off + (size_t)q * p.Cout + o] = v;
        else p.out[(size_t)b * p.out_pitch + (size_t)o * HWo + q] = v;
    }
    if (p.Cout2 > 0) {                                // second head on the same (un-expanded) input: its own depthwise + pointwise, HWC store
        for (int m = 0; m < p.Cexp; m++) for (int oy = 0; oy < p.Ho; oy++) for (int ox = 0; ox < p.Wo; ox++) {
            float s = p.bd_b[m];
            for (int i = 0; i < p.K; i++) for (int j = 0; j < p.K; j++) {
                const int iy = oy * p.S - p.pad + i, ix = ox * p.S - p.pad + j;
                const float x = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? E[(size_t)m * HW + iy * p.W + ix] : 0.f;
                s = fmaf(p.wd_b[(size_t)m * KK + i * p.K + j], x, s);
            }
            Dw[(size_t)m * HWo + oy * p.Wo + ox] = sgx_irb_act(p.act2, s, p.a2c1, p.a2lo, p.a2hi, p.a2c2);
        }
        for (int o = 0; o < p.Cout2; o++) for (int q = 0; q < HWo; q++) {
            float s = p.b2b[o];
            for (int k = 0; k < p.Cexp; k++) s = fmaf(p.w2_b[(size_t)o * p.Cexp + k], Dw[(size_t)k * HWo + q], s);
            p.out2[(size_t)b * p.out2_pitch + (size_t)p.hwc_off2 + (size_t)q * p.Cout2 + o] = s;
        }
    }
}
#endif

// instantiations the planner may pick: (K, S, NT = ceil(Cout / 32), NQ = ceil(Cq / 32), with / without the expand stage)
#define SGX_IRB_INSTANCES(X) \
    X(3, 1, 3, 0, true, true, 0) X(3, 1, 4, 1, true, true, 0) X(5, 1, 5, 2, true, true, 0) X(5, 1, 2, 1, true, false, 0) X(3, 2, 3, 0, true, true, 0) X(5, 2, 2, 1, true, false, 0) \
    X(5, 2, 5, 2, false, true, 0) X(3, 1, 1, 0, false, false, 0) X(3, 1, 3, 0, false, false, 0) X(3, 1, 4, 0, false, false, 0) \
    X(3, 1, 3, 0, false, false, 1) X(3, 1, 4, 0, false, false, 1)
static inline bool sgx_irb_supported(int K, int S, int NT, int NQ, bool expand, bool hswish, int NT2 = 0)
{
#define SGX_IRB_X(K_, S_, NT_, NQ_, E_, H_, N2_) if (K == K_ && S == S_ && NT == NT_ && NQ == NQ_ && expand == E_ && hswish == H_ && NT2 == N2_) return true;
    SGX_IRB_INSTANCES(SGX_IRB_X)
#undef SGX_IRB_X
    return false;
}

#if !defined(SGX_EMU) && defined(SGX_DEBUG_TAPS)
// experiment (round 6): fp32 blob [C][HW] per image -> three bf16 terms in B-operand layout [k16 step][term][half][ldS][8] per image; a thread = (pixel, k16 step, half)
__global__ void __launch_bounds__(256) k_presplit(int C, int HW, int ldS, int nks, int total, const float *in, size_t in_pitch, sgx_u32x4 *out)
{
    const int g = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (g >= total) return;
    const int px = g % HW, r = g / HW, hf = r & 1, s = (r >> 1) % nks, b = (r >> 1) / nks;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { const int k = 16 * s + 8 * hf + j; v[j] = k < C ? in[(size_t)b * in_pitch + (size_t)k * HW + px] : 0.f; }
    const SgxB3 t = sgx_split3x8(v);
    sgx_u32x4 *o = out + (size_t)b * (size_t)(nks * 6 * ldS) + (size_t)(6 * s + hf) * ldS + px;
    o[0] = t.t0; o[(size_t)2 * ldS] = t.t1; o[(size_t)4 * ldS] = t.t2;
}
#endif

#ifndef SGX_IRB_NO_LAUNCH
static inline int sgx_irb_launch(const SgxIrb &p, int batch, sgx_stream_t st)
{
#ifdef SGX_EMU
    for (int b = 0; b < batch; b++) sgx_irb_emu(p, b);
    return SGX_OK;
#else
    const int NT = (p.Cout + 31) / 32, NQ = (p.Cq + 31) / 32, NT2 = (p.Cout2 + 31) / 32;
    const int maxPO = p.nbands > 1 ? p.OH * p.Wo : p.G * p.Ho * p.Wo, ngo = (maxPO + 31) / 32;
    const int maxPI = p.nbands > 1 ? std::min(p.H, (p.OH - 1) * p.S + p.K) * p.W : p.G * p.H * p.W;
    const int nw = std::max(ngo, std::min(12, (maxPI + 31) / 32));               // one wave per output-pixel group; up to 12 waves share the stage-A tiles
    const unsigned grid = p.nbands > 1 ? (unsigned)(batch * p.nbands) : (unsigned)((batch + p.G - 1) / p.G);
    const size_t lds = sgx_irb_lds_bytes(p);
    if (nw > 12 || lds > 160 * 1024) return SGX_ERR_UNSUPPORTED;
    SgxIrb q = p; q.batch = batch;
    { static const int dbg_env = sgx_getenv("SGX_IRB3_DBG") ? atoi(sgx_getenv("SGX_IRB3_DBG")) : 0; q.dbg = dbg_env; }
#ifdef SGX_DEBUG_TAPS
    if (p.gemm == 1) {
        const int nqs = NQ == 0 ? 1 : (NQ == 2 ? 3 : (NT == 2 ? 1 : 2));
        if (NQ > 0 && (p.Cq + 15) / 16 != nqs) return SGX_ERR_UNSUPPORTED;
#define SGX_IRB_X(K_, S_, NT_, NQ_, E_, H_, N2_) if (p.K == K_ && p.S == S_ && NT == NT_ && NQ == NQ_ && (p.has_expand != 0) == E_ && (p.act2 == SGX_EMODE_HSWISH) == H_ && NT2 == N2_) { \
        auto kfn = k_irb3<K_, S_, NT_, NQ_, E_, H_, N2_>; static bool attr = false; \
        if (!attr) { (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * nw), lds, st, q); return SGX_OK; }
        SGX_IRB_INSTANCES(SGX_IRB_X)
#undef SGX_IRB_X
        return SGX_ERR_UNSUPPORTED;
    }
    if (p.gemm == 2 && p.has_expand && p.inS) {
        const int nks = (p.Cin + 15) / 16, total = batch * nks * 2 * p.H * p.W;
        hipLaunchKernelGGL(k_presplit, dim3((total + 255) / 256), dim3(256), 0, st, p.Cin, p.H * p.W, p.ldS, nks, total, p.in, p.in_pitch, (sgx_u32x4 *)p.inS);
    }
    if (p.gemm == 2 && p.has_expand) {
#define SGX_IRB_X(K_, S_, NT_, NQ_, E_, H_, N2_) if (E_ && p.K == K_ && p.S == S_ && NT == NT_ && NQ == NQ_ && (p.act2 == SGX_EMODE_HSWISH) == H_ && NT2 == N2_) { \
        auto kfn = k_irb<K_, S_, NT_, NQ_, E_, H_, N2_, true>; static bool attr = false; \
        if (!attr) { (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * nw), lds, st, q); return SGX_OK; }
        SGX_IRB_INSTANCES(SGX_IRB_X)
#undef SGX_IRB_X
    }
#endif      /* SGX_DEBUG_TAPS: k_irb3 and the bf16x3-expand variant of k_irb */
#define SGX_IRB_X(K_, S_, NT_, NQ_, E_, H_, N2_) if (p.K == K_ && p.S == S_ && NT == NT_ && NQ == NQ_ && (p.has_expand != 0) == E_ && (p.act2 == SGX_EMODE_HSWISH) == H_ && NT2 == N2_) { \
        auto kfn = k_irb<K_, S_, NT_, NQ_, E_, H_, N2_>; static bool attr = false; \
        if (!attr) { (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * nw), lds, st, q); return SGX_OK; }
    SGX_IRB_INSTANCES(SGX_IRB_X)
#undef SGX_IRB_X
    return SGX_ERR_UNSUPPORTED;
#endif
}
#endif
