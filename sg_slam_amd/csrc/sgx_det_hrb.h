// sgx_det_hrb.h — the HIGH-RESOLUTION inverted-residual blocks of the detector backbone (150 x 150 and 75 x 75 maps, 16 - 24 channels in, 16 - 72 expanded; ncnn graph
// mobilenetv3_ssdlite_voc.param layers 588-625, caller Detector2D.cc:34-45) with BOTH pointwise convolutions on the bf16 matrix pipes (bf16x3, sgx_det_bf16.h):
//     pointwise expand Cin -> Cmid (ReLU / Clip)  ->  depthwise K x K stride S (ReLU / Clip)  ->  pointwise project Cmid -> Cout
//     [-> squeeze Cout -> Cq (ReLU) -> excite Cq -> Cout -> hard-sigmoid gate x project output]  [+ residual]
// Round 6.  k_fused_block2 (sgx_det_block.h) runs these blocks as packed fp32 FMAs on the vector pipe: 180 - 270 M wave instructions per 512 frames and launch, 0 % matrix-core
// use, 0.17 - 0.48 of their step rooflines — 2.4 ms of a 10.4 ms forward.  Four fifths of those instructions are the two 1 x 1 convolutions.  Here they become
// v_mfma_f32_32x32x16_bf16 (the only matrix instruction that runs BESIDE the vector work of other waves on gfx950, profiles/r3_ubench_mfma_valu.txt) and the vector pipe keeps
// what has no matrix shape: the depthwise taps, the bf16 splits, the activations.
//
// Work decomposition.  A 256-thread workgroup owns one TOH x TOW output tile of one image.
//   load     wave w takes input-pixel groups g = w, w + 4, ... of the (TOH - 1) S + K by (TOW - 1) S + K input tile (32 pixels each): lane (half, i) loads channels
//            16 s + 8 half + j of pixel 32 g + i straight into the B-operand layout and splits them ONCE into the three bf16 terms (kept in registers over the chunks)
//   per chunk of 32 expanded channels:
//     expand   E = act1(b1 + W1 x X): 6 MFMAs per k16 step and pixel group, accumulator = first MFMA's C operand = the bias rows; rows (r & 3) + 8 (r >> 2) + 4 half of a
//              lane are four channel PAIRS -> eight ds_write_b64 into the pair-interleaved tile E[pair][pixel] (zero outside the image: the depthwise zero padding;
//              stride 2: columns split by parity so that consecutive output pixels read consecutive words)
//     dw       lane = ONE output pixel (64 per wave), channel pairs in turn: K x K ds_read_b64 + v_pk_fma_f32 with the weight pair in scalar registers (wave-uniform:
//              every lane works on the same channels), taps (i, j) ascending from the bias — the fp32 chain of the per-layer kernels
//     operand  the wave's 64 pixels are two 32-pixel groups of the project GEMM.  A lane has channels 16 s .. 16 s + 15 of its pixel; v_permlane32_swap of the registers
//              (channel j, channel 8 + j) hands the lower half-wave's channels 8.. to the upper half and the upper half-wave's channels ..7 to the lower: register j of the
//              result pair IS the B operand (k = 8 half + j) of group 0 resp. group 1.  Split, then 6 MFMAs per group and output tile; accumulators persist over the chunks.
//   tiles of <= 128 output pixels (the stride-2 blocks: their input tiles fill the LDS budget) have two pixel waves only; the other two take every second k16 step of the
//   chunk on the same pixels and the partial accumulators meet in LDS at the end (one more fp32 summation order, as everything in the bf16x3 plan).
//   squeeze-excite (24 -> 72 -> 40 block) on the matrix pipes out of the accumulators as k_irb3: rows -> B layout with four v_permlane32_swap per k16 step.
// Parity: bf16x3 products (dropped terms <= 3 x 2^-24 |a||b|), fp32 accumulation in the MFMA, depthwise exact — the per-step criterion of tests/test_detector.py
// (run_steps_isolated, 4e-6) judges every instantiation; the exact-fp32 plan keeps k_fused_block2.  The emulator build runs THIS source through sgx_lanes.h.
#pragma once
#include "sgx_det_kernels.h"
#include "sgx_det_block.h"
#include "sgx_det_bf16.h"
#include "sgx_lanes.h"

// A/B switches of the round-6 tuning sessions (tools/gpu_session_r6*.sh build variants with -D...)
#ifndef SGX_HRB_PIPE
#define SGX_HRB_PIPE 1       /* 3 x 3 depthwise: weights + taps of the next channel pair requested while the current one multiplies */
#endif
#ifndef SGX_HRB_W1LDS
#define SGX_HRB_W1LDS 1      /* expand weights + bias through LDS, copied once per workgroup */
#endif
#ifndef SGX_HRB_A2PRE
#define SGX_HRB_A2PRE 1      /* project weights requested during the expand stage */
#endif
#ifndef SGX_HRB_RESPRE
#define SGX_HRB_RESPRE 1     /* residual operand requested behind the last depthwise stage */
#endif

struct SgxHrb {
    int Cin, Cmid, Cout, Cq, K, S, pad, H, W, Ho, Wo, TOH, TOW, occ, tiles_x, tiles_y, batch;
    float lo1, hi1, lo2, hi2;
    const float *in; size_t in_pitch; float *out; size_t out_pitch; const float *res; size_t res_pitch;
    const sgx_q4 *w1S; int ld1; const float *b1;             // expand weights as three bf16 terms in the MFMA operand layout [k16 step][term][half][ld1][8] (sgx_split_weights_bf16x3)
    const float *wd2, *bd;                                    // depthwise weights, channel pairs interleaved [Cmid / 2][K * K][2]
    const sgx_q4 *w2S; int ld2; const float *b2;             // project weights, same layout as w1S
    const sgx_q4 *wq1S, *wq2S; int ldq1, ldq2; const float *bq1, *bq2;      // squeeze / excite (Cq != 0)
    float qlo, qhi, gc1, glo, ghi, gc2;                       // squeeze activation; gate = clip(v + gc1, glo, ghi) / gc2
};

template <int CIN, int CMID, int COUT, int K, int S, int TOH, int TOW> struct SgxHrbGeom {
    static constexpr int TIH = (TOH - 1) * S + K, TIW = (TOW - 1) * S + K;
    static constexpr int HALF = (TIW + 1) / 2, TIWP = S == 2 ? 2 * HALF : TIW, ESP = TIH * TIWP + 1;   // E plane of one channel pair, in float2 words (+ one word that takes the stores of lanes past the tile)
    // input pixels are enumerated in the order of their E words (stride 2: row, column parity, half column), so that a pixel group's 32 lanes write 32 consecutive words
    static constexpr int NPI = TIH * TIWP, NGI = (NPI + 31) / 32, GPW = (NGI + 3) / 4;
    static constexpr int NPO = TOH * TOW, NWP = NPO <= 128 ? 2 : 4, KSPLIT = 4 / NWP;             // pixel waves (64 output pixels each), k16 steps of a chunk dealt over KSPLIT waves
    static constexpr int NKS1 = (CIN + 15) / 16, NCH = (CMID + 31) / 32, NKS2 = (CMID + 15) / 16, NT = (COUT + 31) / 32;
    static constexpr int EPAIRS = CMID < 32 ? CMID / 2 : 16;                                       // channel pairs of a chunk held in LDS
    static constexpr int LDS_E0 = EPAIRS * ESP * 8, LDS_RED = KSPLIT == 2 ? NWP * 2 * NT * 16 * 64 * 4 : 0, LDS_E = ((LDS_E0 > LDS_RED ? LDS_E0 : LDS_RED) + 15) & ~15;
    // the expand weights (all chunks, three bf16 terms in operand layout) + bias rows behind the E tile, copied once per workgroup, when two workgroups per CU still fit: a chunk's
    // operands then come from LDS (~100 cycles) instead of a dependent global load at the top of every expand stage
    static constexpr int LDS_W1 = NCH * NKS1 * 6 * 32 * 16, LDS_B1 = NCH * 32 * 4;
    static constexpr bool W1LDS = SGX_HRB_W1LDS && 2 * (LDS_E + LDS_W1 + LDS_B1) <= 160 * 1024;
    static constexpr int LDS_BYTES = LDS_E + (W1LDS ? LDS_W1 + LDS_B1 : 0);
    static_assert(NPO <= 256 && (CMID % 8) == 0 && (CIN % 8) == 0 && (COUT % 8) == 0, "tile / channel constraints");
};

// Output pixel slot o of a tile -> (row, column).  Stride-2 blocks with 8 x 16 tiles deal the rows so that the two 16-lane halves of a 32-lane LDS access group sit FOUR tile rows
// apart: the E words of rows r and r + 4 are 4 * 2 * TIWP = 272 float2 = 32 (mod 64) banks apart, so the 32 lanes of a ds_read_b64 touch 64 distinct banks; with adjacent rows
// (r, r + 1: 8 banks apart) three quarters of every depthwise tap read were two-way bank conflicts (round 6, session B: 27 % of the block's LDS cycles).
#define SGX_HRB_OROW(o_) ((S == 2 && TOW == 16 && TOH == 8) ? vi(((((o_) >> 4) & 1) << 2) + (((o_) >> 5) & 1) + (((o_) >> 6) << 1)) : vi((o_) / TOW))
// NQS: k16 steps of the squeeze width (0 = no squeeze-excite tail); RES: residual tensor added to the output; OCC: waves per SIMD the instantiation is compiled for
// (256-thread workgroups per CU: LDS and registers permitting).  Both activations are ReLU / Clip(0, hi) (lo1 = lo2 = 0: the planner checks).
template <int CIN, int CMID, int COUT, int K, int S, int TOH, int TOW, int NQS, bool RES, int OCC>
SGX_KERNEL_OCC(256, OCC) k_hrb(SgxHrb p)
{
    typedef SgxHrbGeom<CIN, CMID, COUT, K, S, TOH, TOW> G;
    constexpr int KK = K * K, NT = G::NT;
    constexpr bool A2PRE = SGX_HRB_A2PRE && K == 3 && NT == 1;      // the 5 x 5 / two-tile block has no registers to spare (measured: 8 -> 59 spilled registers, 0.51 -> 0.82 ms)
    SGX_DYN_LDS(smem);
    sgx_f2 *Es = (sgx_f2 *)smem;
    float *Red = (float *)smem;
    sgx_q4 *W1L = (sgx_q4 *)(smem + G::LDS_E);                  // [chunk][k16 step][term][half][32 rows] (W1LDS)
    float *B1L = (float *)(smem + G::LDS_E + G::LDS_W1);       // [chunk][32 rows], zero past Cmid
    // Cin = 16 n + 8 (the 24-channel blocks): the last k16 step carries eight channels, i.e. only the lower half-wave's operand slots (k = 0..7) meet non-zero weight rows.
    // Two pixel groups SHARE the registers of that step — group 2 u in the lower half-wave, group 2 u + 1 in the upper — and the odd group multiplies with the weight operand
    // read with the half-wave index flipped (real rows in the upper half's slots k = 8..15, the zero rows below): 12 registers per group pair instead of 24.
    constexpr bool SH = (CIN % 16) == 8;
    constexpr int NXS = SH ? G::NKS1 - 1 : G::NKS1, NPAIR = (G::GPW + 1) / 2;      // full k16 steps per group; group pairs
    SGX_WPRIV_DECL(VB3, xs, G::GPW * NXS + (SH ? NPAIR : 1));      // split input operands: [group][full step], then [pair] of the shared step
    SGX_WPRIV_DECL(vi, eix, G::GPW);                          // E word of the group's pixel (lanes past the tile: the plane's spare word), bit 30: outside the image
    SGX_WPRIV_DECL(vf16, acc, 2 * NT);                        // project accumulators: [pixel group 0 / 1][output tile]
    // operands requested one phase ahead of their use (a dependent global load costs 1 - 2 us here; round 6 session B: the first version spent 80 % of its wave cycles waiting)
    SGX_WPRIV_DECL(vu4, a2r, 2 * NT * 3);                     // project weights of the chunk's two k16 steps, requested while its expand stage runs
    SGX_WPRIV_DECL(vf, resr, RES ? 2 * NT * 16 : 1);          // residual operand, requested ahead of the last depthwise stage
    int tile, b;
    sgx_xcd_order((int)blockIdx.x, p.tiles_x * p.tiles_y, p.batch, &b, &tile);
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int oy0 = ty * TOH, ox0 = tx * TOW, iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
    const float *X = p.in + (size_t)b * p.in_pitch;
    const unsigned plane4 = (unsigned)(p.H * p.W) * 4u;

    // ---- load: the wave's input-pixel groups -> split B operands
    SGX_WAVES_BEGIN(w)
    SGX_WPRIV_BIND(xs, w); SGX_WPRIV_BIND(eix, w); SGX_WPRIV_BIND(acc, w);
    const vi lane = v_lane(), l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int gi = 0; gi < G::GPW; gi++) {
        const int g = gi * 4 + w;
        const vi q = g * 32 + l31;                                   // = the pixel's E word
        const vi qc = v_min(q, vi(G::NPI - 1)), ry = qc / G::TIWP, rem = qc - ry * G::TIWP, par = S == 2 ? vi(rem / G::HALF) : vi(0), rx = S == 2 ? vi(2 * (rem - par * G::HALF) + par) : rem;
        const vi iy = iy0 + ry, ix = ix0 + rx;
        const vb valid = (q < G::NPI) & (rx < G::TIW);
        const vb inside = valid & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
        eix[gi] = v_seli(valid, q, vi(G::ESP - 1)) | v_seli(inside, vi(0), vi(1 << 30));
        const vu xoffh = v_u(v_seli(inside, iy * p.W + ix, vi(0))) * 4u + v_u(half) * (8u * plane4);      // the upper half-wave starts eight channels on
#pragma unroll
        for (int s = 0; s < NXS; s++) {
            vf xv[8];
#pragma unroll
            for (int j = 0; j < 8; j++)          // channel 16 s + 8 half + j: wave-uniform plane in the base pointer, lane offset = pixel (+ 8 planes)
                xv[j] = v_sel(inside, v_ld((const float *)((const char *)X + (size_t)(16 * s + j) * plane4), xoffh), vf(0.f));
            xs[gi * NXS + s] = v_split3x8(xv);
        }
    }
    if (SH) {
#pragma unroll
        for (int u = 0; u < NPAIR; u++) {        // the shared last step: lane (half, i) holds channels Cin - 8 + j of pixel i of group 2 u + half
            const vi q = ((2 * u + half) * 4 + w) * 32 + l31;
            const vi qc = v_min(q, vi(G::NPI - 1)), ry = qc / G::TIWP, rem = qc - ry * G::TIWP, par = S == 2 ? vi(rem / G::HALF) : vi(0), rx = S == 2 ? vi(2 * (rem - par * G::HALF) + par) : rem;
            const vi iy = iy0 + ry, ix = ix0 + rx;
            const vb inside = (q < G::NPI) & (rx < G::TIW) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
            const vu xo = v_u(v_seli(inside, iy * p.W + ix, vi(0))) * 4u;
            vf xv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) xv[j] = v_sel(inside, v_ld((const float *)((const char *)X + (size_t)(CIN - 8 + j) * plane4), xo), vf(0.f));
            xs[G::GPW * NXS + u] = v_split3x8(xv);
        }
    }
    if (G::W1LDS) {                                               // the workgroup's copy of the expand weights and bias rows (visible behind the first barrier)
#pragma unroll
        for (int i0 = 0; i0 < G::LDS_W1 / 16; i0 += 256) {
            const vi i = i0 + w * 64 + lane, ic = v_min(i, vi(G::LDS_W1 / 16 - 1));
            const vi row = ic & 31, blk = ic >> 5, cch = blk / (G::NKS1 * 6), sth = blk - cch * (G::NKS1 * 6);      // block = (chunk, k16 step, term, half)
            v_lds_stq(W1L, ic, v_ldq(p.w1S, sth * p.ld1 + 32 * cch + row), i < G::LDS_W1 / 16);
        }
        if (w == 0) {
#pragma unroll
            for (int i0 = 0; i0 < G::NCH * 32; i0 += 64) {
                const vi i = i0 + lane;
                v_lds_st(B1L, v_min(i, vi(G::NCH * 32 - 1)), v_sel(i < CMID, v_ld(p.b1, v_u(v_min(i, vi(CMID - 1))) * 4u), vf(0.f)));
            }
        }
    }
    // accumulators start from the project bias (rows past Cout: zero)
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const vi row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            const vf bz = v_sel(row < COUT, v_ld(p.b2, v_u(v_min(row, vi(COUT - 1))) * 4u), vf(0.f));
            acc[t][r] = w / G::NWP == 0 ? bz : vf(0.f); acc[NT + t][r] = acc[t][r];
        }
    SGX_WAVES_END
    if (G::W1LDS) SGX_SYNC();                                    // the weight copy is complete

    for (int c = 0; c < G::NCH; c++) {
        // ---- expand chunk c (expanded channels 32 c .. 32 c + 31) into the E tile
        SGX_WAVES_BEGIN(w)
        SGX_WPRIV_BIND(xs, w); SGX_WPRIV_BIND(eix, w); SGX_WPRIV_BIND(a2r, w);
        const vi lane = v_lane(), l31 = lane & 31, half = lane >> 5;
        vu4 a1r[G::NKS1 * 3 + 3]; vf16 b1v;                       // this chunk's expand operands: lane (half, i) = row 32 c + i, k = 8 half + j (the shared step also with the half index flipped)
#pragma unroll
        for (int s_ = 0; s_ < G::NKS1; s_++)
#pragma unroll
            for (int t_ = 0; t_ < 3; t_++)
                a1r[s_ * 3 + t_] = G::W1LDS ? v_lds_ldq(W1L, (((c * G::NKS1 + s_) * 3 + t_) * 2 + half) * 32 + l31) : v_ldq(p.w1S, ((s_ * 3 + t_) * 2 + half) * p.ld1 + 32 * c + l31);
        if (SH) {
#pragma unroll
            for (int t_ = 0; t_ < 3; t_++)
                a1r[G::NKS1 * 3 + t_] = G::W1LDS ? v_lds_ldq(W1L, (((c * G::NKS1 + G::NKS1 - 1) * 3 + t_) * 2 + (1 - half)) * 32 + l31) : v_ldq(p.w1S, (((G::NKS1 - 1) * 3 + t_) * 2 + (1 - half)) * p.ld1 + 32 * c + l31);
        }
#pragma unroll
        for (int r_ = 0; r_ < 16; r_++) {
            const vi row_ = 32 * c + (r_ & 3) + 8 * (r_ >> 2) + 4 * half;
            b1v[r_] = G::W1LDS ? v_lds_ld(B1L, row_) : v_sel(row_ < CMID, v_ld(p.b1, v_u(v_min(row_, vi(CMID - 1))) * 4u), vf(0.f));
        }
#pragma unroll
        for (int sl = 0; sl < 2; sl++)                            // project weights of the steps this wave will take behind the barrier
            if (A2PRE && (CMID % 32 == 0 || 16 * (2 * c + sl) < CMID) && (G::KSPLIT == 1 || sl == w / G::NWP)) {
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int q = 0; q < 3; q++) a2r[(sl * NT + t) * 3 + q] = v_ldq(p.w2S, (((2 * c + sl) * 3 + q) * 2 + half) * p.ld2 + 32 * t + l31);
            }
#pragma unroll
        for (int gi = 0; gi < G::GPW; gi++) {
            if (gi * 4 + w < G::NGI) {                            // wave-uniform
                vf16 e = b1v;
#pragma unroll
                for (int s = 0; s < NXS; s++) e = v_mfma3(a1r[s * 3], a1r[s * 3 + 1], a1r[s * 3 + 2], xs[gi * NXS + s], e);
                if (SH) e = (gi & 1) ? v_mfma3(a1r[G::NKS1 * 3], a1r[G::NKS1 * 3 + 1], a1r[G::NKS1 * 3 + 2], xs[G::GPW * NXS + gi / 2], e)
                                     : v_mfma3(a1r[(G::NKS1 - 1) * 3], a1r[(G::NKS1 - 1) * 3 + 1], a1r[(G::NKS1 - 1) * 3 + 2], xs[G::GPW * NXS + gi / 2], e);
                const vi ei = eix[gi];
                const vi li = ei & 0xFFFFFF;
                const vf hil = v_sel((ei & (1 << 30)) != 0, vf(0.f), vf(p.hi1));     // zero outside the image (the depthwise convolution pads ITS input): ReLU / Clip(0, hi) as ONE v_med3 with a per-lane upper bound
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    if (CMID % 32 == 0 || 32 * c + 8 * m < CMID) {    // rows 8 m + 4 half + 0..3 = channel pairs 4 m + 2 half, + 1 of the chunk
                        const vi pl = 4 * m + 2 * half;
                        v_lds_st2(Es, pl * G::ESP + li, v_clipv(e[4 * m], 0.f, hil), v_clipv(e[4 * m + 1], 0.f, hil), vb(true));
                        v_lds_st2(Es, (pl + 1) * G::ESP + li, v_clipv(e[4 * m + 2], 0.f, hil), v_clipv(e[4 * m + 3], 0.f, hil), vb(true));
                    }
                }
            }
        }
        SGX_WAVES_END
        SGX_SYNC();
        // ---- depthwise + project on the chunk's k16 steps
        SGX_WAVES_BEGIN(w)
        SGX_WPRIV_BIND(acc, w); SGX_WPRIV_BIND(a2r, w); SGX_WPRIV_BIND(resr, w);
        const vi lane = v_lane(), l31 = lane & 31, half = lane >> 5;
        const int pw = w % G::NWP, ks = w / G::NWP;
        const vi o = v_min(pw * 64 + lane, vi(G::NPO - 1)), oy = SGX_HRB_OROW(o), ox = o - (o / TOW) * TOW;
        const vi ebase = (oy * S) * G::TIWP + ox;                  // tap (a, c): + a TIWP + (S == 2 ? (c & 1) HALF + (c >> 1) : c)
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            const int s = 2 * c + sl;
            if ((CMID % 32 == 0 || 16 * s < CMID) && (G::KSPLIT == 1 || sl == ks)) {  // wave-uniform
                if (!A2PRE) {
#pragma unroll
                    for (int t = 0; t < NT; t++)
#pragma unroll
                        for (int q = 0; q < 3; q++) a2r[(sl * NT + t) * 3 + q] = v_ldq(p.w2S, ((s * 3 + q) * 2 + half) * p.ld2 + 32 * t + l31);
                }
                vf dX[8], dY[8];
                if (K == 3 && SGX_HRB_PIPE) {
                    // 3 x 3: the weights of a channel pair are scalar loads, and scalar loads return out of order — the only wait that covers one is "everything", which also drains
                    // the LDS queue.  So a pair's weights AND taps are requested while the PREVIOUS pair multiplies (two register sets): the first multiply-add of pair pp takes the
                    // one wait, then pair pp + 1 is requested, then the other eight multiply-adds run on registers that have arrived.
                    sgx_f2 wk[2][KK], bk[2]; vf2 tp[2][KK];
#define SGX_HRB_ISSUE(pp_, u_) do { const int gp_ = 8 * s + (pp_); const sgx_f2 *wt_ = (const sgx_f2 *)p.wd2 + (size_t)gp_ * KK;                                   \
                        bk[u_] = sgx_mk2(p.bd[2 * gp_], p.bd[2 * gp_ + 1]);                                                                                         \
                        _Pragma("unroll") for (int t_ = 0; t_ < KK; t_++) wk[u_][t_] = wt_[t_];                                                                    \
                        _Pragma("unroll") for (int t_ = 0; t_ < KK; t_++)                                                                                          \
                            tp[u_][t_] = v_lds_ld2(Es, ebase + ((8 * sl + (pp_)) * G::ESP + (t_ / K) * G::TIWP + (S == 2 ? ((t_ % K) & 1) * G::HALF + ((t_ % K) >> 1) : (t_ % K)))); } while (0)
                    SGX_HRB_ISSUE(0, 0);
#pragma unroll
                    for (int pp = 0; pp < 8; pp++) {
                        const int gp = 8 * s + pp, u = pp & 1;
                        vf2 d = v_mk2(vf(0.f), vf(0.f));
                        if (CMID % 16 == 0 || 2 * gp < CMID) {        // wave-uniform
                            vf2 sv = v_fma2_w(wk[u][0], tp[u][0], v_mk2(vf(bk[u].x), vf(bk[u].y)));
                            SGX_SCHED_FENCE();
                            if (pp + 1 < 8 && (CMID % 16 == 0 || 2 * (gp + 1) < CMID)) SGX_HRB_ISSUE(pp + 1, u ^ 1);
                            SGX_SCHED_FENCE();
#pragma unroll
                            for (int t = 1; t < KK; t++) sv = v_fma2_w(wk[u][t], tp[u][t], sv);
                            d = v_mk2(v_clip(v_x(sv), p.lo2, p.hi2), v_clip(v_y(sv), p.lo2, p.hi2));
                        }
                        if (pp < 4) { dX[2 * pp] = v_x(d); dX[2 * pp + 1] = v_y(d); } else { dY[2 * (pp - 4)] = v_x(d); dY[2 * (pp - 4) + 1] = v_y(d); }
                    }
#undef SGX_HRB_ISSUE
                } else {
#pragma unroll
                for (int pp = 0; pp < 8; pp++) {
                    const int gp = 8 * s + pp;                    // channel pair of the block
                    vf2 d = v_mk2(vf(0.f), vf(0.f));
                    if (CMID % 16 == 0 || 2 * gp < CMID) {
                        const sgx_f2 *wt = (const sgx_f2 *)p.wd2 + (size_t)gp * KK;
                        vf2 sv = v_mk2(vf(p.bd[2 * gp]), vf(p.bd[2 * gp + 1]));
#pragma unroll
                        for (int a = 0; a < K; a++)
#pragma unroll
                            for (int cc = 0; cc < K; cc++)
                                sv = v_fma2_w(wt[a * K + cc], v_lds_ld2(Es, ebase + ((8 * sl + pp) * G::ESP + a * G::TIWP + (S == 2 ? (cc & 1) * G::HALF + (cc >> 1) : cc))), sv);
                        d = v_mk2(v_clip(v_x(sv), p.lo2, p.hi2), v_clip(v_y(sv), p.lo2, p.hi2));
                    }
                    if (pp < 4) { dX[2 * pp] = v_x(d); dX[2 * pp + 1] = v_y(d); } else { dY[2 * (pp - 4)] = v_x(d); dY[2 * (pp - 4) + 1] = v_y(d); }
                }
                }
                // residual operand of the wave's two pixel groups (as the store below addresses them), requested behind the wave's last depthwise stage: in flight during the split, the
                // matrix products and the closing barriers
                if (RES && SGX_HRB_RESPRE && c == G::NCH - 1 && ks == 0 && sl == (G::KSPLIT == 2 || !(CMID % 32 == 0 || 16 * (2 * c + 1) < CMID) ? 0 : 1)) {
        #pragma unroll
                    for (int g = 0; g < 2; g++) {
                        const vi o_ = pw * 64 + 32 * g + l31, oc_ = v_min(o_, vi(G::NPO - 1)), oy_ = SGX_HRB_OROW(oc_), ox_ = oc_ - (oc_ / TOW) * TOW, gy_ = oy0 + oy_, gx_ = ox0 + ox_;
                        const vb live_ = (o_ < G::NPO) & (gy_ < p.Ho) & (gx_ < p.Wo);
                        const vu pix4_ = v_u(v_seli(live_, gy_ * p.Wo + gx_, vi(0))) * 4u + v_u(4 * half) * ((unsigned)(p.Ho * p.Wo) * 4u);
        #pragma unroll
                        for (int t = 0; t < NT; t++)
        #pragma unroll
                            for (int r = 0; r < 16; r++) {
                                const int rb = 32 * t + (r & 3) + 8 * (r >> 2);
                                if (rb < COUT) resr[(g * NT + t) * 16 + r] = v_sel(live_, v_ld(p.res + (size_t)b * p.res_pitch, pix4_ + (unsigned)rb * ((unsigned)(p.Ho * p.Wo) * 4u)), vf(0.f));
                            }
                    }
                }
                vf g0[8], g1[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v_swap32(dX[j], dY[j], g0[j], g1[j]);
                const VB3 b0 = v_split3x8(g0), b1 = v_split3x8(g1);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    acc[t] = v_mfma3(a2r[(sl * NT + t) * 3], a2r[(sl * NT + t) * 3 + 1], a2r[(sl * NT + t) * 3 + 2], b0, acc[t]);
                    acc[NT + t] = v_mfma3(a2r[(sl * NT + t) * 3], a2r[(sl * NT + t) * 3 + 1], a2r[(sl * NT + t) * 3 + 2], b1, acc[NT + t]);
                }
            }
        }
        SGX_WAVES_END
        SGX_SYNC();
    }

    // ---- the k-split halves meet: waves of the second k slice park their accumulators in LDS (the E tile is dead), the pixel waves add them
    if (G::KSPLIT == 2) {
        SGX_WAVES_BEGIN(w)
        SGX_WPRIV_BIND(acc, w);
        const vi lane = v_lane();
        if (w / G::NWP == 1) {
#pragma unroll
            for (int u = 0; u < 2 * NT; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) v_lds_st(Red, (((w % G::NWP) * 2 * NT + u) * 16 + r) * 64 + lane, acc[u][r]);
        }
        SGX_WAVES_END
        SGX_SYNC();
    }
    SGX_WAVES_BEGIN(w)
    SGX_WPRIV_BIND(acc, w); SGX_WPRIV_BIND(resr, w);
    if (w / G::NWP != 0) SGX_WAVE_EXIT();
    const vi lane = v_lane(), l31 = lane & 31, half = lane >> 5;
    const int pw = w % G::NWP;
    if (G::KSPLIT == 2) {
#pragma unroll
        for (int u = 0; u < 2 * NT; u++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[u][r] = acc[u][r] + v_lds_ld(Red, ((pw * 2 * NT + u) * 16 + r) * 64 + lane);
    }
    // ---- squeeze-excite gate on the matrix pipes, out of the accumulators
    if (NQS > 0) {
        const float rcg = 1.0f / p.gc2;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            vf16 qa;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const vi row = (r & 3) + 8 * (r >> 2) + 4 * half;
                qa[r] = v_sel(row < p.Cq, v_ld(p.bq1, v_u(v_min(row, vi(p.Cq - 1))) * 4u), vf(0.f));
            }
#pragma unroll
            for (int k2 = 0; k2 < (COUT + 15) / 16; k2++) {       // k16 step k2 of the squeeze = rows 16 k2 .. 16 k2 + 15 of the project output: register quads 8 q .. of tile k2 >> 1
                const int t = k2 >> 1, q = k2 & 1;
                vf bv[8];
#pragma unroll
                for (int i = 0; i < 4; i++) v_swap32(acc[g * NT + t][8 * q + i], acc[g * NT + t][8 * q + 4 + i], bv[i], bv[4 + i]);
                const VB3 bq = v_split3x8(bv);
                vu4 aq[3];
#pragma unroll
                for (int m = 0; m < 3; m++) aq[m] = v_ldq(p.wq1S, ((k2 * 3 + m) * 2 + half) * p.ldq1 + l31);
                qa = v_mfma3(aq[0], aq[1], aq[2], bq, qa);
            }
            VB3 qb[NQS > 0 ? NQS : 1];
#pragma unroll
            for (int k2 = 0; k2 < NQS; k2++) {
                vf hv[16], bv[8];
#pragma unroll
                for (int r = 0; r < 8; r++) hv[r] = v_clip(qa[8 * k2 + r], p.qlo, p.qhi);
#pragma unroll
                for (int i = 0; i < 4; i++) v_swap32(hv[i], hv[4 + i], bv[i], bv[4 + i]);
                qb[k2] = v_split3x8(bv);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
                vf16 ga;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const vi row = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                    ga[r] = v_sel(row < COUT, v_ld(p.bq2, v_u(v_min(row, vi(COUT - 1))) * 4u), vf(0.f));
                }
#pragma unroll
                for (int k2 = 0; k2 < NQS; k2++) {
                    vu4 ae[3];
#pragma unroll
                    for (int m = 0; m < 3; m++) ae[m] = v_ldq(p.wq2S, ((k2 * 3 + m) * 2 + half) * p.ldq2 + 32 * t + l31);
                    ga = v_mfma3(ae[0], ae[1], ae[2], qb[k2], ga);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {                    // [ADD c][CLIP][DIV c][MUL project output], as sgx_epi_mode<SGX_EMODE_GATE>
                    // the division as q0 = u rc, q = fma(fma(-q0, c2, u), rc, q0) (rc = RN(1 / c2)): the correctly rounded quotient unless |u| < 2^-123 (sgx_irb_act_fast, sgx_det_irb.h)
                    vf u_ = ga[r] + p.gc1; u_ = v_clip(u_, p.glo, p.ghi);
                    const vf q0 = u_ * rcg; u_ = v_fma(v_fma(-1.f * q0, vf(p.gc2), u_), vf(rcg), q0);
                    acc[g * NT + t][r] = u_ * acc[g * NT + t][r];
                }
            }
        }
    }
    // ---- store (+ residual): pixel group g of the wave = output pixels 64 pw + 32 g + i, rows of tile t = output channels
    float *Y = p.out + (size_t)b * p.out_pitch;
    const unsigned oplane4 = (unsigned)(p.Ho * p.Wo) * 4u;
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const vi o = pw * 64 + 32 * g + l31, oc = v_min(o, vi(G::NPO - 1)), oy = SGX_HRB_OROW(oc), ox = oc - (oc / TOW) * TOW, gy = oy0 + oy, gx = ox0 + ox;
        const vb live = (o < G::NPO) & (gy < p.Ho) & (gx < p.Wo);
        const vu pix4 = v_u(v_seli(live, gy * p.Wo + gx, vi(0))) * 4u + v_u(4 * half) * oplane4;
        if (RES && !SGX_HRB_RESPRE) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (32 * t + (r & 3) + 8 * (r >> 2) < COUT) resr[(g * NT + t) * 16 + r] = v_sel(live, v_ld(p.res + (size_t)b * p.res_pitch, pix4 + (unsigned)(32 * t + (r & 3) + 8 * (r >> 2)) * oplane4), vf(0.f));
        }
        if (RES) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (32 * t + (r & 3) + 8 * (r >> 2) < COUT) acc[g * NT + t][r] = acc[g * NT + t][r] + resr[(g * NT + t) * 16 + r];      // COUT is a multiple of 8: row + 4 half < COUT too
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rb = 32 * t + (r & 3) + 8 * (r >> 2);
                if (rb < COUT) v_st(Y, pix4 + (unsigned)rb * oplane4, acc[g * NT + t][r], live);
            }
    }
    SGX_WAVES_END
}

// ---- dispatch: (Cin, Cmid, Cout, K, S, Cq, residual) -> instantiation and its tile -----------------------------------------------------------------------------
#define SGX_HRB_INSTANCES(X)                                                                  \
    X(16, 16, 16, 3, 1, 16, 16, 0, true, 4)   /* 588+591+594: 150 x 150, + residual          */  \
    X(16, 64, 24, 3, 2, 8, 16, 0, false, 2)   /* 597+600+603: 150 -> 75                      */  \
    X(24, 72, 24, 3, 1, 16, 16, 0, true, 2)   /* 605+608+611: 75 x 75, + residual            */  \
    X(24, 72, 40, 5, 2, 5, 19, 1, false, 2)   /* 614+617+620+622+625: 75 -> 38, squeeze-excite */ \
    SGX_HRB_ALTERNATIVES(X)
// (measured and dropped, session r6c: the two 40 -> 120 -> 40 blocks at 38 x 38 (5 x 5, squeeze-excite, + residual) as X(40, 120, 40, 5, 1, 6, 19, 1, true, 2): 0.65 ms each against 0.52 for
// their four per-layer launches — two output tiles of accumulators + 25 taps in flight do not fit 256 registers: 31 - 78 spilled)
// alternative tiles / occupancies of the same blocks (tap build: SGX_HRB_PICK=n takes the n-th instantiation that fits a block; the product plans the first)
#ifdef SGX_DEBUG_TAPS
#define SGX_HRB_ALTERNATIVES(X) \
    X(16, 16, 16, 3, 1, 16, 16, 0, true, 5) X(16, 64, 24, 3, 2, 7, 16, 0, false, 2) X(24, 72, 24, 3, 1, 16, 16, 0, true, 3) \
    X(16, 16, 16, 3, 1, 16, 16, 0, true, 6) X(24, 72, 24, 3, 1, 8, 16, 0, true, 3) X(24, 72, 40, 5, 2, 4, 19, 1, false, 2) X(24, 72, 40, 5, 2, 6, 19, 1, false, 2)
#else
#define SGX_HRB_ALTERNATIVES(X)
#endif
static inline bool sgx_hrb_variant(int cin, int cmid, int cout, int k, int s, int cq, bool res, float lo1, float lo2, int *toh, int *tow, int *occ)
{
    if (lo1 != 0.f || lo2 != 0.f) return false;
    static const int pick_env = sgx_getenv("SGX_HRB_PICK") ? atoi(sgx_getenv("SGX_HRB_PICK")) : 0;
    int seen = 0; bool found = false;
#define SGX_HRB_X(CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_) \
    if (cin == CIN_ && cmid == CMID_ && cout == COUT_ && k == K_ && s == S_ && ((cq + 15) / 16) == NQS_ && res == RES_) { if (!found || seen <= pick_env) { *toh = TOH_; *tow = TOW_; *occ = OCC_; found = true; } seen++; }
    SGX_HRB_INSTANCES(SGX_HRB_X)
#undef SGX_HRB_X
    return found;
}
static inline int sgx_hrb_launch(const SgxHrb &p0, int batch, sgx_stream_t st)
{
    SgxHrb p = p0; p.batch = batch;
    const unsigned grid = (unsigned)(p.tiles_x * p.tiles_y * batch);
#ifndef SGX_EMU
#define SGX_HRB_X(CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_) \
    if (p.Cin == CIN_ && p.Cmid == CMID_ && p.Cout == COUT_ && p.K == K_ && p.S == S_ && ((p.Cq + 15) / 16) == NQS_ && (p.res != nullptr) == RES_ && p.TOH == TOH_ && p.TOW == TOW_ && p.occ == OCC_) { \
        auto kfn = k_hrb<CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_>; constexpr int lds = SgxHrbGeom<CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_>::LDS_BYTES; static bool attr = false; \
        if (!attr) { (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, p); return SGX_OK; }
#else
#define SGX_HRB_X(CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_) \
    if (p.Cin == CIN_ && p.Cmid == CMID_ && p.Cout == COUT_ && p.K == K_ && p.S == S_ && ((p.Cq + 15) / 16) == NQS_ && (p.res != nullptr) == RES_ && p.TOH == TOH_ && p.TOW == TOW_ && p.occ == OCC_) { \
        auto kfn = k_hrb<CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_>; SGX_LAUNCH(kfn, dim3(grid), dim3(256), st, p); return SGX_OK; }
#endif
    SGX_HRB_INSTANCES(SGX_HRB_X)
#undef SGX_HRB_X
    return SGX_ERR_INVALID;
}
