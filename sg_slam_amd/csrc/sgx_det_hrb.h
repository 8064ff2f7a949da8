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

struct SgxHrb {
    int Cin, Cmid, Cout, Cq, K, S, pad, H, W, Ho, Wo, TOH, TOW, tiles_x, tiles_y, batch;
    float lo1, hi1, lo2, hi2;
    const float *in; size_t in_pitch; float *out; size_t out_pitch; const float *res; size_t res_pitch;
    const sgx_q4 *w1S; int ld1; const float *b1;             // expand weights as three bf16 terms in the MFMA operand layout [k16 step][term][half][ld1][8] (sgx_split_weights_bf16x3)
    const float *wd2, *bd;                                    // depthwise weights, channel pairs interleaved [Cmid / 2][K * K][2]
    const sgx_q4 *w2S; int ld2; const float *b2;             // project weights, same layout as w1S
    const sgx_q4 *wq1S, *wq2S; int ldq1, ldq2; const float *bq1, *bq2;      // squeeze / excite (Cq != 0)
    float qlo, qhi, gc1, glo, ghi, gc2;                       // squeeze activation; gate = clip(v + gc1, glo, ghi) / gc2
};

template <int CIN, int CMID, int COUT, int K, int S, int TOH, int TOW> struct SgxHrbGeom {
    static constexpr int TIH = (TOH - 1) * S + K, TIW = (TOW - 1) * S + K, NPI = TIH * TIW, NGI = (NPI + 31) / 32, GPW = (NGI + 3) / 4;
    static constexpr int NPO = TOH * TOW, NWP = NPO <= 128 ? 2 : 4, KSPLIT = 4 / NWP;             // pixel waves (64 output pixels each), k16 steps of a chunk dealt over KSPLIT waves
    static constexpr int HALF = (TIW + 1) / 2, TIWP = S == 2 ? 2 * HALF : TIW, ESP = TIH * TIWP + 1;   // E plane of one channel pair, in float2 words (+ one word that takes the stores of lanes past the tile)
    static constexpr int NKS1 = (CIN + 15) / 16, NCH = (CMID + 31) / 32, NKS2 = (CMID + 15) / 16, NT = (COUT + 31) / 32;
    static constexpr int EPAIRS = CMID < 32 ? CMID / 2 : 16;                                       // channel pairs of a chunk held in LDS
    static constexpr int LDS_E = EPAIRS * ESP * 8, LDS_RED = KSPLIT == 2 ? NWP * 2 * NT * 16 * 64 * 4 : 0, LDS_BYTES = LDS_E > LDS_RED ? LDS_E : LDS_RED;
    static_assert(NPO <= 256 && (CMID % 8) == 0 && (CIN % 8) == 0 && (COUT % 8) == 0, "tile / channel constraints");
};

// NQS: k16 steps of the squeeze width (0 = no squeeze-excite tail); RES: residual tensor added to the output; OCC: waves per SIMD the instantiation is compiled for
// (256-thread workgroups per CU: LDS and registers permitting).  Both activations are ReLU / Clip(0, hi) (lo1 = lo2 = 0: the planner checks).
template <int CIN, int CMID, int COUT, int K, int S, int TOH, int TOW, int NQS, bool RES, int OCC>
SGX_KERNEL_OCC(256, OCC) k_hrb(SgxHrb p)
{
    typedef SgxHrbGeom<CIN, CMID, COUT, K, S, TOH, TOW> G;
    constexpr int KK = K * K, NT = G::NT;
    SGX_DYN_LDS(smem);
    sgx_f2 *Es = (sgx_f2 *)smem;
    float *Red = (float *)smem;
    SGX_WPRIV_DECL(VB3, xs, G::GPW * G::NKS1);               // split input operands of the wave's pixel groups
    SGX_WPRIV_DECL(vi, eix, G::GPW);                          // E word of the group's pixel (lanes past the tile: the plane's spare word), bit 30: outside the image
    SGX_WPRIV_DECL(vf16, acc, 2 * NT);                        // project accumulators: [pixel group 0 / 1][output tile]
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
        const vi q = g * 32 + l31;
        const vb valid = q < G::NPI;
        const vi qc = v_min(q, vi(G::NPI - 1)), ry = qc / G::TIW, rx = qc - ry * G::TIW, iy = iy0 + ry, ix = ix0 + rx;
        const vb inside = valid & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
        const vi li = S == 2 ? vi(ry * G::TIWP + (rx & 1) * G::HALF + (rx >> 1)) : vi(ry * G::TIWP + rx);
        eix[gi] = v_seli(valid, li, vi(G::ESP - 1)) | v_seli(inside, vi(0), vi(1 << 30));
        const vu xoff = v_u(v_seli(inside, iy * p.W + ix, vi(0))) * 4u, xoffh = xoff + v_u(half) * (8u * plane4);      // the upper half-wave starts eight channels on
#pragma unroll
        for (int s = 0; s < G::NKS1; s++) {
            vf xv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                // channel 16 s + 8 half + j: wave-uniform plane in the base pointer, lane offset = pixel (+ 8 planes).  Channels past Cin (second step of a 24-channel
                // input, upper half-wave) are clamped: they meet zero weight rows
                const vf ld = 16 * s + 8 + j < CIN ? v_ld((const float *)((const char *)X + (size_t)(16 * s + j) * plane4), xoffh)
                                                   : v_ld(X, xoff + v_u(v_min(vi(16 * s + 8 * half + j), vi(CIN - 1))) * plane4);
                xv[j] = v_sel(inside, ld, vf(0.f));
            }
            xs[gi * G::NKS1 + s] = v_split3x8(xv);
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

    for (int c = 0; c < G::NCH; c++) {
        // ---- expand chunk c (expanded channels 32 c .. 32 c + 31) into the E tile
        SGX_WAVES_BEGIN(w)
        SGX_WPRIV_BIND(xs, w); SGX_WPRIV_BIND(eix, w);
        const vi lane = v_lane(), l31 = lane & 31, half = lane >> 5;
        vu4 a1[G::NKS1][3];
#pragma unroll
        for (int s = 0; s < G::NKS1; s++)
#pragma unroll
            for (int t = 0; t < 3; t++) a1[s][t] = v_ldq(p.w1S, ((s * 3 + t) * 2 + half) * p.ld1 + 32 * c + l31);
        vf16 bias1;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const vi row = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * half;
            bias1[r] = v_sel(row < CMID, v_ld(p.b1, v_u(v_min(row, vi(CMID - 1))) * 4u), vf(0.f));
        }
#pragma unroll
        for (int gi = 0; gi < G::GPW; gi++) {
            if (gi * 4 + w < G::NGI) {                            // wave-uniform
                vf16 e = bias1;
#pragma unroll
                for (int s = 0; s < G::NKS1; s++) e = v_mfma3(a1[s][0], a1[s][1], a1[s][2], xs[gi * G::NKS1 + s], e);
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
        SGX_WPRIV_BIND(acc, w);
        const vi lane = v_lane(), l31 = lane & 31, half = lane >> 5;
        const int pw = w % G::NWP, ks = w / G::NWP;
        const vi o = v_min(pw * 64 + lane, vi(G::NPO - 1)), oy = o / TOW, ox = o - oy * TOW;
        const vi ebase = (oy * S) * G::TIWP + ox;                  // tap (a, c): + a TIWP + (S == 2 ? (c & 1) HALF + (c >> 1) : c)
#pragma unroll
        for (int sl = 0; sl < 2; sl++) {
            const int s = 2 * c + sl;
            if ((CMID % 32 == 0 || 16 * s < CMID) && (G::KSPLIT == 1 || sl == ks)) {  // wave-uniform
                vu4 a2[NT][3];
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int q = 0; q < 3; q++) a2[t][q] = v_ldq(p.w2S, ((s * 3 + q) * 2 + half) * p.ld2 + 32 * t + l31);
                vf dX[8], dY[8];
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
                vf g0[8], g1[8];
#pragma unroll
                for (int j = 0; j < 8; j++) v_swap32(dX[j], dY[j], g0[j], g1[j]);
                const VB3 b0 = v_split3x8(g0), b1 = v_split3x8(g1);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    acc[t] = v_mfma3(a2[t][0], a2[t][1], a2[t][2], b0, acc[t]);
                    acc[NT + t] = v_mfma3(a2[t][0], a2[t][1], a2[t][2], b1, acc[NT + t]);
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
    SGX_WPRIV_BIND(acc, w);
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
                    vf u_ = ga[r] + p.gc1; u_ = v_clip(u_, p.glo, p.ghi); u_ = u_ / p.gc2;
                    acc[g * NT + t][r] = u_ * acc[g * NT + t][r];
                }
            }
        }
    }
    // ---- store (+ residual): pixel group g of the wave = output pixels 64 pw + 32 g + i, rows of tile t = output channels
    float *Y = p.out + (size_t)b * p.out_pitch;
    const float *R = RES ? p.res + (size_t)b * p.res_pitch : nullptr;
    const unsigned oplane4 = (unsigned)(p.Ho * p.Wo) * 4u;
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const vi o = pw * 64 + 32 * g + l31, oc = v_min(o, vi(G::NPO - 1)), oy = oc / TOW, ox = oc - oy * TOW, gy = oy0 + oy, gx = ox0 + ox;
        const vb live = (o < G::NPO) & (gy < p.Ho) & (gx < p.Wo);
        const vu pix4 = v_u(v_seli(live, gy * p.Wo + gx, vi(0))) * 4u + v_u(4 * half) * oplane4;
        if (RES) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rb = 32 * t + (r & 3) + 8 * (r >> 2);
                    if (rb < COUT) {                              // COUT is a multiple of 8: rb + 4 half < COUT too
                        const vf rv = v_ld(R, pix4 + (unsigned)rb * oplane4);
                        acc[g * NT + t][r] = acc[g * NT + t][r] + v_sel(live, rv, vf(0.f));
                    }
                }
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
    X(24, 72, 40, 5, 2, 5, 19, 1, false, 2)   /* 614+617+620+622+625: 75 -> 38, squeeze-excite */
static inline bool sgx_hrb_variant(int cin, int cmid, int cout, int k, int s, int cq, bool res, float lo1, float lo2, int *toh, int *tow)
{
    if (lo1 != 0.f || lo2 != 0.f) return false;
#define SGX_HRB_X(CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_) \
    if (cin == CIN_ && cmid == CMID_ && cout == COUT_ && k == K_ && s == S_ && ((cq + 15) / 16) == NQS_ && res == RES_) { *toh = TOH_; *tow = TOW_; return true; }
    SGX_HRB_INSTANCES(SGX_HRB_X)
#undef SGX_HRB_X
    return false;
}
static inline int sgx_hrb_launch(const SgxHrb &p0, int batch, sgx_stream_t st)
{
    SgxHrb p = p0; p.batch = batch;
    const unsigned grid = (unsigned)(p.tiles_x * p.tiles_y * batch);
#ifndef SGX_EMU
#define SGX_HRB_X(CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_) \
    if (p.Cin == CIN_ && p.Cmid == CMID_ && p.Cout == COUT_ && p.K == K_ && p.S == S_ && ((p.Cq + 15) / 16) == NQS_ && (p.res != nullptr) == RES_ && p.TOH == TOH_ && p.TOW == TOW_) { \
        auto kfn = k_hrb<CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_>; constexpr int lds = SgxHrbGeom<CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_>::LDS_BYTES; static bool attr = false; \
        if (!attr) { (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; } \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, st, p); return SGX_OK; }
#else
#define SGX_HRB_X(CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_) \
    if (p.Cin == CIN_ && p.Cmid == CMID_ && p.Cout == COUT_ && p.K == K_ && p.S == S_ && ((p.Cq + 15) / 16) == NQS_ && (p.res != nullptr) == RES_ && p.TOH == TOH_ && p.TOW == TOW_) { \
        auto kfn = k_hrb<CIN_, CMID_, COUT_, K_, S_, TOH_, TOW_, NQS_, RES_, OCC_>; SGX_LAUNCH(kfn, dim3(grid), dim3(256), st, p); return SGX_OK; }
#endif
    SGX_HRB_INSTANCES(SGX_HRB_X)
#undef SGX_HRB_X
    return SGX_ERR_INVALID;
}
