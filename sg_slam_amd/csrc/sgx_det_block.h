// sgx_det_block.h — fused inverted-residual block of the detector backbone:
//     pointwise expand (Cin -> Cmid, ReLU / Clip)  ->  depthwise K x K, stride s (ReLU / Clip)  ->  pointwise project (Cmid -> Cout) [+ residual tensor]
// as ONE kernel per output tile.  The unfused plan writes the expanded Cmid-channel tensor to HBM twice and reads it twice (at 150 x 150 x 64 that is
// 5.8 MB per image per trip); here it only ever exists as one 32-channel chunk of one tile in LDS.
//   phase A  E[32 cm][tile pixels] = act1(b1 + W1 x X)          v_mfma_f32_32x32x2_f32, accumulators start from the bias, k ascending (exact fp32 FMA chain)
//   phase B  D[32 cm][out pixels]  = act2(bd + Wd * E)          VALU, taps (i, j) ascending with fmaf, taps in the zero padding add an exact zero
//   phase C  O[co][out pixels]    += W2 x D                      same MFMA, accumulators persist over the Cmid chunks (k keeps ascending)
// Arithmetic and operation order are those of the one-kernel-per-layer plan (k_conv_pw / k_conv_kxk): the results are bit-identical to it.
#pragma once
#include "sgx_det_kernels.h"

struct SgxFusedBlk {
    int Cin, Cmid, Cout, K, stride, pad, H, W, Ho, Wo;
    int TOH, TOW, tiles_x, tiles_y, TIH, TIW, NPI, NPO;      // output tile, tiles per image, input tile, pixel counts rounded up to 32
    int CMR, ES;                                              // rows of a chunk held in LDS (min(32, Cmid)), row stride of the E tile (NPI + 4: phase B reads stay conflict-free)
    int dbg;                                                   // tuning tap (SGX_FB_DBG bit mask): 1 skip input staging, 2 skip phase A, 4 skip phase B, 8 skip phase C, 16 skip the store
    unsigned m_tiw, m_tow, m_npo, m_kk;                        // ceil(2^32 / d) for the divisions by TIW, TOW, TOH*TOW, K*K (sgx_fastdiv)
    float lo1, hi1, lo2, hi2;
    const float *in; size_t in_pitch;
    const float *w1, *b1, *wd, *bd, *w2, *b2;                 // original ncnn layouts: w1 [Cmid][Cin], wd [Cmid][K*K], w2 [Cout][Cmid]
    float *out; size_t out_pitch;
    const float *res; size_t res_pitch;                        // residual (BinaryOp add behind the project convolution) or NULL
    int v2;                                                    // != 0: k_fused_block2 instantiation (VALU-only, high-resolution blocks), tile variant in the low bits
    const float *w2t; int ldw2;                                // project weights transposed [Cmid][ldw2] (the pointwise kernel's copy)
    const float *wd2;                                          // depthwise weights, channel pairs interleaved [Cmid / 2][K * K][2]
    // squeeze-excite tail (k_fused_block2 only, Cq != 0): gate = clip(bq2 + Wq2 x clip(bq1 + Wq1 x y, qlo, qhi) + gc1, glo, ghi) / gc2, output = gate * y [+ residual]
    int Cq; float qlo, qhi, gc1, glo, ghi, gc2;
    const float *wq1, *bq1, *wq2, *bq2;                        // original layouts: wq1 [Cq][Cout], wq2 [Cout][Cq]
    const float *wq1p, *wq2p;                                  // pair-interleaved copies: wq1p [Cq / 2][Cout][2] = (wq1[2j][k], wq1[2j + 1][k]), wq2p [Cq][Cout / 2][2] = (wq2[2c][j], wq2[2c + 1][j])    // round 6: != 0 runs the block as k_hrb (sgx_det_hrb.h: both pointwise convolutions as bf16x3 on the matrix pipes; bf16x3 plan only) — split weights of the expand, project,
    // squeeze and excite convolutions in the MFMA operand layout (sgx_split_weights_bf16x3) with their leading dimensions
    int hrb, hrb_occ; const void *w1S, *w2S, *wq1S, *wq2S; int ld1S, ld2S, ldq1S, ldq2S;
};
struct SgxFb2Se { const float *wq1p, *bq1, *wq2p, *bq2; float qlo, qhi, gc1, glo, ghi, gc2; };
#define SGX_FB_CM 32                                           /* expanded channels per chunk = one MFMA row block */
static inline size_t sgx_fb_lds_floats(const SgxFusedBlk &p)
{
    const int coP = ((p.Cout + 31) / 32) * 32;
    return (size_t)p.Cin * p.NPI + (size_t)p.Cin * SGX_FB_CM + (size_t)p.CMR * p.ES + (size_t)SGX_FB_CM * p.K * p.K + 2 * SGX_FB_CM + (size_t)p.CMR * p.NPO +
           (size_t)p.CMR * coP;
}

#ifdef SGX_DEBUG_TAPS      /* the first fused-block kernel (fp32 MFMA, 13.7 against 10.9 ms per forward): superseded, tap build only (sgx_det_debug_set_block_fusion / SGX_DET_BLOCK_FUSION) */
SGX_KERNEL(256) k_fused_block(SgxFusedBlk p)
{
    SGX_DYN_LDS(smem);
    const int coP = ((p.Cout + 31) / 32) * 32, cosh = coP == 32 ? 5 : 6, KK = p.K * p.K;
    float *Xs = (float *)smem;                                 // [Cin][NPI]
    float *W1s = Xs + (size_t)p.Cin * p.NPI;                   // [Cin][32]      (k, cm)
    float *Es = W1s + (size_t)p.Cin * SGX_FB_CM;               // [CMR][ES]
    float *Wds = Es + (size_t)p.CMR * p.ES;                    // [32][K*K]
    float *b1s = Wds + (size_t)SGX_FB_CM * KK, *bds = b1s + SGX_FB_CM;
    float *Ds = bds + SGX_FB_CM;                               // [CMR][NPO]     (k = cm, pixel)
    float *W2s = Ds + (size_t)p.CMR * p.NPO;                   // [CMR][coP]     (k = cm, co)
    const int tile = (int)blockIdx.x % (p.tiles_x * p.tiles_y), b = (int)blockIdx.x / (p.tiles_x * p.tiles_y);
    const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const int oy0 = ty * p.TOH, ox0 = tx * p.TOW;              // first output pixel of the tile
    const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
    const int npi = p.TIH * p.TIW, npo = p.TOH * p.TOW;
    const float *X = p.in + (size_t)b * p.in_pitch;
    const int nchunks = (p.Cmid + SGX_FB_CM - 1) / SGX_FB_CM;
    const int NBI = p.NPI / 32, NBO = p.NPO / 32, NCB = coP / 32;

    // ---- stage the input tile (all Cin channels; zeros outside the image and in the round-up tail), clear the D tail
    SGX_THREADS_BEGIN(tid)
    for (int q = tid; q < ((p.dbg & 1) ? 0 : p.NPI); q += 256) {
        const int ry = (int)sgx_fastdiv((unsigned)q, p.m_tiw), rx = q - ry * p.TIW, iy = iy0 + ry, ix = ix0 + rx;
        const bool ok = q < npi && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const float *src = X + (size_t)(ok ? iy : 0) * p.W + (ok ? ix : 0);
        const size_t plane = (size_t)p.H * p.W;
        for (int c0 = 0; c0 < p.Cin; c0 += 8) {                    // eight independent loads in flight per thread before the LDS stores (latency, not instruction count, bounds this phase)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(size_t)min(c0 + u, p.Cin - 1) * plane];
#pragma unroll
            for (int u = 0; u < 8; u++) if (c0 + u < p.Cin) Xs[(size_t)(c0 + u) * p.NPI + q] = ok ? v[u] : 0.f;
        }
    }
    for (int i = tid; i < p.CMR * p.NPO; i += 256) Ds[i] = 0.f;
    SGX_THREADS_END
#ifndef SGX_EMU
    const int tid_ = (int)threadIdx.x, wave = tid_ >> 6, lane = tid_ & 63, l31 = lane & 31, lh = lane >> 5;
    sgx_f32x16 accC[2];                                        // phase C accumulators of this wave: pairs q = wave, wave + 4 of (co block, pixel block)
#endif
    for (int ch = 0; ch < nchunks; ch++) {
        const int cm0 = ch * SGX_FB_CM, ncm = min(SGX_FB_CM, p.Cmid - cm0);
        SGX_SYNC();
        // ---- this chunk's weights (zero rows beyond Cmid)
        SGX_THREADS_BEGIN(tid)
        {   // all weight loads of the chunk are issued before the first LDS store
            const int n1 = p.Cin * SGX_FB_CM, n2 = ncm * KK, n3 = ncm * coP;
            float a1[8], a2[4], a3[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = tid + 256 * u, k = i >> 5, m = i & 31; a1[u] = (i < n1 && m < ncm) ? p.w1[(size_t)(cm0 + m) * p.Cin + k] : 0.f; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = tid + 256 * u; a2[u] = i < n2 ? p.wd[(size_t)cm0 * KK + i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = tid + 256 * u, m = i >> cosh, co = i & (coP - 1); a3[u] = (i < n3 && co < p.Cout) ? p.w2[(size_t)co * p.Cmid + cm0 + m] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = tid + 256 * u; if (i < n1) W1s[i] = a1[u]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = tid + 256 * u; if (i < n2) Wds[i] = a2[u]; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int i = tid + 256 * u; if (i < n3) W2s[i] = a3[u]; }
        }
        if (tid < SGX_FB_CM) { b1s[tid] = tid < ncm ? p.b1[cm0 + tid] : 0.f; bds[tid] = tid < ncm ? p.bd[cm0 + tid] : 0.f; }
        SGX_THREADS_END
        SGX_SYNC();
        // ---- phase A: expand on the matrix cores, activation, zero outside the image (the depthwise convolution pads ITS input with zeros)
#ifndef SGX_EMU
        for (int pb = wave; pb < ((p.dbg & 2) ? 0 : NBI); pb += 4) {
            sgx_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = b1s[(r & 3) + 8 * (r >> 2) + 4 * lh];
            for (int k = 0; k < p.Cin; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W1s[(k + lh) * SGX_FB_CM + l31], Xs[(size_t)(k + lh) * p.NPI + pb * 32 + l31], acc, 0, 0, 0);
            const int q = pb * 32 + l31, ry = (int)sgx_fastdiv((unsigned)q, p.m_tiw), rx = q - ry * p.TIW, iy = iy0 + ry, ix = ix0 + rx;
            const bool inside = q < npi && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < ncm) Es[(size_t)row * p.ES + q] = inside ? fminf(fmaxf(acc[r], p.lo1), p.hi1) : 0.f;
            }
        }
#else
        SGX_THREADS_BEGIN(tid)
        for (int m = 0; m < ncm; m++) for (int q = tid; q < p.NPI; q += 256) {
            const int ry = q / p.TIW, rx = q - ry * p.TIW, iy = iy0 + ry, ix = ix0 + rx;
            const bool inside = q < npi && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            float s = b1s[m];
            for (int k = 0; k < p.Cin; k++) s = fmaf(W1s[k * SGX_FB_CM + m], Xs[(size_t)k * p.NPI + q], s);
            Es[(size_t)m * p.ES + q] = inside ? fminf(fmaxf(s, p.lo1), p.hi1) : 0.f;
        }
        SGX_THREADS_END
#endif
        SGX_SYNC();
        // ---- phase B: depthwise K x K on the chunk (consecutive threads -> consecutive output pixels of one channel)
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < ((p.dbg & 4) ? 0 : ncm * npo); i += 256) {
            const int m = (int)sgx_fastdiv((unsigned)i, p.m_npo), o = i - m * npo, oy = (int)sgx_fastdiv((unsigned)o, p.m_tow), ox = o - oy * p.TOW;
            const float *e = Es + (size_t)m * p.ES + (oy * p.stride) * p.TIW + ox * p.stride, *w = Wds + m * KK;
            float s = bds[m];
            if (p.K == 3) {
#pragma unroll
                for (int a = 0; a < 3; a++) {
#pragma unroll
                    for (int c = 0; c < 3; c++) s = fmaf(w[a * 3 + c], e[a * p.TIW + c], s);
                }
            } else {
                for (int a = 0; a < p.K; a++) for (int c = 0; c < p.K; c++) s = fmaf(w[a * p.K + c], e[a * p.TIW + c], s);
            }
            Ds[(size_t)m * p.NPO + o] = fminf(fmaxf(s, p.lo2), p.hi2);
        }
        SGX_THREADS_END
        SGX_SYNC();
        // ---- phase C: project, accumulators persist over the chunks
#ifndef SGX_EMU
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int q = wave + 4 * u;
            if (q < NCB * NBO) {
                const int cb = q / NBO, pb = q - cb * NBO;
                if (ch == 0) {
#pragma unroll
                    for (int r = 0; r < 16; r++) { const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh; accC[u][r] = co < p.Cout ? p.b2[co] : 0.f; }
                }
                for (int k = 0; k < ((p.dbg & 8) ? 0 : ncm); k += 2) accC[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(W2s[(k + lh) * coP + cb * 32 + l31], Ds[(size_t)(k + lh) * p.NPO + pb * 32 + l31], accC[u], 0, 0, 0);
            }
        }
#endif
    }
    // ---- store (+ residual)
#ifndef SGX_EMU
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int q = wave + 4 * u;
        if (q < NCB * NBO) {
            const int cb = q / NBO, pb = q - cb * NBO;
            const int o = pb * 32 + l31, oy = o / p.TOW, ox = o - oy * p.TOW, gy = oy0 + oy, gx = ox0 + ox;
            if (o < npo && gy < p.Ho && gx < p.Wo && !(p.dbg & 16)) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (co < p.Cout) {
                        const size_t off = ((size_t)co * p.Ho + gy) * p.Wo + gx;
                        float v = accC[u][r];
                        if (p.res) v = v + p.res[(size_t)b * p.res_pitch + off];
                        p.out[(size_t)b * p.out_pitch + off] = v;
                    }
                }
            }
        }
    }
#else
    // kernel-logic emulator: the project convolution as a scalar fmaf chain over ALL expanded channels (recomputes E / D per chunk exactly like the device path;
    // the chunk loop above only exercised the staging code in this build), same order: bias, then cm ascending
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) {
        static thread_local float accs[64 * 1024];
        for (int i = 0; i < p.Cout * npo; i++) accs[i] = p.b2[i / npo];
        for (int ch = 0; ch < nchunks; ch++) {
            const int cm0 = ch * SGX_FB_CM, ncm = min(SGX_FB_CM, p.Cmid - cm0);
            for (int m = 0; m < ncm; m++) {
                for (int q = 0; q < npi; q++) {
                    const int ry = q / p.TIW, rx = q - ry * p.TIW, iy = iy0 + ry, ix = ix0 + rx;
                    float s = p.b1[cm0 + m];
                    for (int k = 0; k < p.Cin; k++) s = fmaf(p.w1[(size_t)(cm0 + m) * p.Cin + k], Xs[(size_t)k * p.NPI + q], s);
                    Es[q] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? fminf(fmaxf(s, p.lo1), p.hi1) : 0.f;
                }
                for (int o = 0; o < npo; o++) {
                    const int oy = o / p.TOW, ox = o - oy * p.TOW;
                    float s = p.bd[cm0 + m];
                    for (int a = 0; a < p.K; a++) for (int c = 0; c < p.K; c++) s = fmaf(p.wd[(size_t)(cm0 + m) * KK + a * p.K + c], Es[(oy * p.stride + a) * p.TIW + ox * p.stride + c], s);
                    const float d = fminf(fmaxf(s, p.lo2), p.hi2);
                    for (int co = 0; co < p.Cout; co++) accs[co * npo + o] = fmaf(p.w2[(size_t)co * p.Cmid + cm0 + m], d, accs[co * npo + o]);
                }
            }
        }
        for (int co = 0; co < p.Cout; co++) for (int o = 0; o < npo; o++) {
            const int oy = o / p.TOW, ox = o - oy * p.TOW, gy = oy0 + oy, gx = ox0 + ox;
            if (gy < p.Ho && gx < p.Wo) {
                const size_t off = ((size_t)co * p.Ho + gy) * p.Wo + gx;
                float v = accs[co * npo + o];
                if (p.res) v = v + p.res[(size_t)b * p.res_pitch + off];
                p.out[(size_t)b * p.out_pitch + off] = v;
            }
        }
    }
    SGX_THREADS_END
#endif
}
#endif      /* SGX_DEBUG_TAPS */

// ---------------------------------------------------------------------------------------------
// k_fused_block2 — the same inverted-residual block for the HIGH-RESOLUTION, FEW-CHANNEL blocks at the head of the backbone (150 x 150 and 75 x 75, Cin / Cout <= 24),
// where the one-kernel-per-layer plan is pure activation traffic (the expanded tensor of the 16 -> 64 block is 5.8 MB per image, written once and read once)
// and the matrix cores have nothing to amortise (K = 16).  Everything is VALU fmaf in the unfused kernels' order, so the result is bit-identical to them:
//   once      x[slot][k]      the thread's input pixels (tile + halo dealt round-robin over 128 threads), all Cin channels, straight from global memory into registers
//   per chunk of CM expanded channels:
//     phase A  E[m][pixel] = act1(b1 + sum_k w1[m][k] x[k])   k ascending; zero outside the image (the depthwise convolution pads ITS input)        -> LDS, one chunk only
//     phase B  d = act2(bd + sum_taps wd[m][t] E[m][tap])      taps (i, j) ascending; one output pixel per thread and slot
//     phase C  acc[co] = fmaf(w2[co][m], d, acc[co])           accumulators start from b2 and persist over the chunks (m keeps ascending)
//   store      acc (+ residual tensor)
// Weights are read with scalar loads (uniform addresses, __restrict__): w1 [Cmid][CIN], wd [Cmid][K*K], w2t [Cmid][ldw2] (the transposed copy the pointwise kernel uses).
// For stride 2 the E tile is stored with its columns split by parity, so the 16 lanes of an output row read consecutive words for every tap.
// ---------------------------------------------------------------------------------------------
#define SGX_FB2_THREADS 128
#ifndef SGX_EMU
#define SGX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SGX_SCHED_FENCE() do { } while (0)
#endif
// (the packed multiply-add helpers sgx_f2 / sgx_fma2_* live in sgx_det_kernels.h since round 6: the stem uses them too)
template <int CIN, int COUT, int K, int S, int TOH, int TOW, int CM> struct SgxFb2Geom {
    static constexpr int TIH = (TOH - 1) * S + K, TIW = (TOW - 1) * S + K, NPI = TIH * TIW, NPO = TOH * TOW;
    static constexpr int SLOTS_IN = (NPI + SGX_FB2_THREADS - 1) / SGX_FB2_THREADS, SLOTS_OUT = (NPO + SGX_FB2_THREADS - 1) / SGX_FB2_THREADS;
    static constexpr int PAIRS_IN = SLOTS_IN / 2, ODD_IN = SLOTS_IN & 1;
    static constexpr int HALF = (TIW + 1) / 2;
    static constexpr int TIWP = S == 2 ? 2 * HALF + 1 : TIW + 1;          // row pitch of the E tile in channel-pair words (stride 2: two half rows)
    static constexpr int ES = TIH * TIWP;                                  // pixels per channel pair
    static_assert((CM & 1) == 0 && (COUT & 1) == 0 && (CIN & 1) == 0, "channels are processed in pairs");
};

// wd2: depthwise weights with the two channels of a pair interleaved, [Cmid / 2][K * K][2] (built once per plan, sgx_det.cpp)
// RES: 0 = no residual operand, 1 = fetched before the chunk loop (hides its latency, costs COUT registers), 2 = fetched before the store; UA: 2 = phase A on channel pairs (8-byte LDS stores), 1 = one channel per step, 3 = one channel per step unrolled by two
// NQ: channels of the squeeze-excite tail behind the project convolution (0 = none).  The block's output pixel has all COUT channels in one thread's accumulators, and this
// network's squeeze-excite has no pooling (1 x 1 convolutions on the feature map itself), so the tail is two small matrix-vector products per thread on wave-uniform weights:
// the same fmaf chains, k ascending from the bias, as the pointwise kernels of the per-layer plan.
// PIPE = 1: software-pipelined weight fetch.  The weights of a step are scalar loads (SMEM returns out of order, so the only wait is "all of them"), and a block with 40 output
// channels and a 5 x 5 depthwise needs 130 weight registers per channel pair — more than the scalar file holds, the compiler then parks them in VGPR lanes (v_writelane /
// v_readlane: 108 extra VALU instructions around 65 FMAs, measured).  With PIPE the loop body is cut into regions by scheduling fences; every region first issues the loads the
// NEXT region needs and then computes on registers loaded one region earlier, so at most one region's weights plus the next one's are live and the load latency sits behind
// a region's FMAs.
template <int CIN, int COUT, int K, int S, int TOH, int TOW, int CM, int RES, int UA, int NQ = 0, int PIPE = 0>
SGX_KERNEL(SGX_FB2_THREADS) k_fused_block2(int Cmid, int H, int W, int Ho, int Wo, int pad, int tiles_x, int tiles_y, float lo1, float hi1, float lo2, float hi2,
                                           const float *__restrict__ in, size_t in_pitch, const float *__restrict__ w1, const float *__restrict__ b1,
                                           const float *__restrict__ wd2, const float *__restrict__ bd, const float *__restrict__ w2t, int ldw2, const float *__restrict__ b2,
                                           float *__restrict__ out, size_t out_pitch, const float *__restrict__ res, size_t res_pitch, SgxFb2Se se)
{
    typedef SgxFb2Geom<CIN, COUT, K, S, TOH, TOW, CM> G;
    constexpr int UNR_A = UA == 3 ? 2 : 1;
    SGX_LDS sgx_f2 Es[(CM / 2) * G::ES];                                                       // E tile of the chunk: [channel pair][pixel] -> (E[2p][pixel], E[2p + 1][pixel])
    SGX_PRIV_DECL(sgx_f2, x2, (G::PAIRS_IN ? G::PAIRS_IN : 1) * CIN, SGX_FB2_THREADS);        // input pixels of slots (2p, 2p + 1), all Cin channels
    SGX_PRIV_DECL(float, x1, CIN, SGX_FB2_THREADS);                                            // the odd last slot
    SGX_PRIV_DECL(sgx_f2, acc, G::SLOTS_OUT * (COUT / 2), SGX_FB2_THREADS);                    // output channels (2c, 2c + 1)
    SGX_PRIV_DECL(int, eidx, G::SLOTS_IN, SGX_FB2_THREADS);                    // pixel index inside the E tile (-1: slot beyond the tile); bit 30 set = pixel outside the image
    SGX_PRIV_DECL(float, rsd, RES == 1 ? G::SLOTS_OUT * COUT : 1, SGX_FB2_THREADS);           // residual operand of the thread's output pixels, fetched up front (its latency hides behind the whole block)
    int tile, b;
    sgx_xcd_order((int)blockIdx.x, tiles_x * tiles_y, (int)gridDim.x / (tiles_x * tiles_y), &b, &tile);      // a frame's tiles share halo rows and cache lines: one XCD (one L2) per frame
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * TOH, ox0 = tx * TOW, iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;
    const float *X = in + (size_t)b * in_pitch;
    const size_t plane = (size_t)H * W;

    SGX_THREADS_BEGIN(tid)
    SGX_PRIV_BIND(x2, tid); SGX_PRIV_BIND(x1, tid); SGX_PRIV_BIND(acc, tid); SGX_PRIV_BIND(eidx, tid); SGX_PRIV_BIND(rsd, tid);
    if (RES == 1) {
#pragma unroll
        for (int jo = 0; jo < G::SLOTS_OUT; jo++) {
            const int o = tid + SGX_FB2_THREADS * jo, oy = o / TOW, ox = o - oy * TOW, gy = oy0 + oy, gx = ox0 + ox;
            const bool live = o < G::NPO && gy < Ho && gx < Wo;
            const float *rp = res + (live ? (size_t)b * res_pitch + (size_t)gy * Wo + gx : 0);
#pragma unroll
            for (int co = 0; co < COUT; co++) { const float ld = rp[live ? (size_t)co * Ho * Wo : 0]; rsd[jo * COUT + co] = live ? ld : 0.f; }
        }
    }
#pragma unroll
    for (int j = 0; j < G::SLOTS_IN; j++) {
        const int q = tid + SGX_FB2_THREADS * j, ry = q / G::TIW, rx = q - ry * G::TIW, iy = iy0 + ry, ix = ix0 + rx;
        const bool inside = q < G::NPI && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float *src = X + (size_t)(inside ? iy : 0) * W + (inside ? ix : 0);          // every lane loads (a valid address), the select follows: no divergent branches
#pragma unroll
        for (int k = 0; k < CIN; k++) {
            const float ld = src[(size_t)k * plane];
            const float v = inside ? ld : 0.f;
            if (j < 2 * G::PAIRS_IN) { if (j & 1) x2[(j >> 1) * CIN + k].y = v; else x2[(j >> 1) * CIN + k].x = v; }
            else x1[k] = v;
        }
        const int li = S == 2 ? ry * G::TIWP + (rx & 1) * G::HALF + (rx >> 1) : ry * G::TIWP + rx;
        eidx[j] = q < G::NPI ? (li | (inside ? 0 : (1 << 30))) : -1;
    }
#pragma unroll
    for (int jo = 0; jo < G::SLOTS_OUT; jo++) {
#pragma unroll
        for (int cp = 0; cp < COUT / 2; cp++) acc[jo * (COUT / 2) + cp] = sgx_mk2(b2[2 * cp], b2[2 * cp + 1]);
    }
    SGX_THREADS_END

    for (int cm0 = 0; cm0 < Cmid; cm0 += CM) {
        // ---- phase A: expand this chunk for the thread's input pixels: two pixels per packed FMA (the weight broadcast to both halves), two expanded channels per step so
        //      that a pixel's pair (E[m], E[m + 1]) leaves as one 8-byte LDS store into the pair-interleaved tile
        if (PIPE) {
        SGX_THREADS_BEGIN(tid)
        SGX_PRIV_BIND(x2, tid); SGX_PRIV_BIND(x1, tid); SGX_PRIV_BIND(eidx, tid);
        float *Ef = (float *)Es;
        sgx_f2 wk[CIN / 2], wn[CIN / 2]; float bias, biasn;
        {
            const sgx_f2 *wr = (const sgx_f2 *)(w1 + (size_t)cm0 * CIN);
#pragma unroll
            for (int k = 0; k < CIN / 2; k++) wk[k] = wr[k];
            bias = b1[cm0];
        }
#pragma unroll 1
        for (int m = 0; m < CM; m++) {
            {   // the next channel's weights (the last step reloads its own: harmless)
                const int mn = cm0 + (m + 1 < CM ? m + 1 : m);
                const sgx_f2 *wr = (const sgx_f2 *)(w1 + (size_t)mn * CIN);
#pragma unroll
                for (int k = 0; k < CIN / 2; k++) wn[k] = wr[k];
                biasn = b1[mn];
            }
            SGX_SCHED_FENCE();
            float *Em = Ef + (size_t)(m >> 1) * G::ES * 2 + (m & 1);
            sgx_f2 sp[G::PAIRS_IN ? G::PAIRS_IN : 1];
#pragma unroll
            for (int jp = 0; jp < G::PAIRS_IN; jp++) sp[jp] = sgx_mk2(bias, bias);
            float s1 = bias;
#pragma unroll
            for (int k = 0; k < CIN / 2; k++) {
#pragma unroll
                for (int jp = 0; jp < G::PAIRS_IN; jp++) sp[jp] = sgx_fma2_wlo(wk[k], x2[jp * CIN + 2 * k], sp[jp]);
                if (G::ODD_IN) s1 = fmaf(wk[k].x, x1[2 * k], s1);
#pragma unroll
                for (int jp = 0; jp < G::PAIRS_IN; jp++) sp[jp] = sgx_fma2_whi(wk[k], x2[jp * CIN + 2 * k + 1], sp[jp]);
                if (G::ODD_IN) s1 = fmaf(wk[k].y, x1[2 * k + 1], s1);
            }
#pragma unroll
            for (int jp = 0; jp < G::PAIRS_IN; jp++) {
                const sgx_f2 sv = sp[jp];
                const int e0 = eidx[2 * jp], e1 = eidx[2 * jp + 1];
                if (e0 >= 0) Em[(e0 & 0xFFFFFF) * 2] = (e0 & (1 << 30)) ? 0.f : sgx_clipf(sv.x, lo1, hi1);
                if (e1 >= 0) Em[(e1 & 0xFFFFFF) * 2] = (e1 & (1 << 30)) ? 0.f : sgx_clipf(sv.y, lo1, hi1);
            }
            if (G::ODD_IN) {
                const int e = eidx[G::SLOTS_IN - 1];
                if (e >= 0) Em[(e & 0xFFFFFF) * 2] = (e & (1 << 30)) ? 0.f : sgx_clipf(s1, lo1, hi1);
            }
            SGX_SCHED_FENCE();
#pragma unroll
            for (int k = 0; k < CIN / 2; k++) wk[k] = wn[k];
            bias = biasn;
        }
        SGX_THREADS_END
        } else if (UA == 2) {
        SGX_THREADS_BEGIN(tid)
        SGX_PRIV_BIND(x2, tid); SGX_PRIV_BIND(x1, tid); SGX_PRIV_BIND(eidx, tid);
#pragma unroll 1
        for (int m = 0; m < CM; m += 2) {
            const sgx_f2 *wr0 = (const sgx_f2 *)(w1 + (size_t)(cm0 + m) * CIN), *wr1 = (const sgx_f2 *)(w1 + (size_t)(cm0 + m + 1) * CIN);
            const float bias0 = b1[cm0 + m], bias1 = b1[cm0 + m + 1];
            sgx_f2 wk0[CIN / 2], wk1[CIN / 2];
#pragma unroll
            for (int k = 0; k < CIN / 2; k++) { wk0[k] = wr0[k]; wk1[k] = wr1[k]; }
            sgx_f2 *Em = Es + (size_t)(m >> 1) * G::ES;
            sgx_f2 sp0[G::PAIRS_IN ? G::PAIRS_IN : 1], sp1[G::PAIRS_IN ? G::PAIRS_IN : 1];
#pragma unroll
            for (int jp = 0; jp < G::PAIRS_IN; jp++) { sp0[jp] = sgx_mk2(bias0, bias0); sp1[jp] = sgx_mk2(bias1, bias1); }
#pragma unroll
            for (int k = 0; k < CIN / 2; k++) {                      // independent chains (pixel pairs x two channels) interleaved
#pragma unroll
                for (int jp = 0; jp < G::PAIRS_IN; jp++) { sp0[jp] = sgx_fma2_wlo(wk0[k], x2[jp * CIN + 2 * k], sp0[jp]); sp1[jp] = sgx_fma2_wlo(wk1[k], x2[jp * CIN + 2 * k], sp1[jp]); }
#pragma unroll
                for (int jp = 0; jp < G::PAIRS_IN; jp++) { sp0[jp] = sgx_fma2_whi(wk0[k], x2[jp * CIN + 2 * k + 1], sp0[jp]); sp1[jp] = sgx_fma2_whi(wk1[k], x2[jp * CIN + 2 * k + 1], sp1[jp]); }
            }
#pragma unroll
            for (int jp = 0; jp < G::PAIRS_IN; jp++) {
                const int e0 = eidx[2 * jp], e1 = eidx[2 * jp + 1];
                if (e0 >= 0) Em[e0 & 0xFFFFFF] = (e0 & (1 << 30)) ? sgx_mk2(0.f, 0.f) : sgx_mk2(sgx_clipf(sp0[jp].x, lo1, hi1), sgx_clipf(sp1[jp].x, lo1, hi1));
                if (e1 >= 0) Em[e1 & 0xFFFFFF] = (e1 & (1 << 30)) ? sgx_mk2(0.f, 0.f) : sgx_mk2(sgx_clipf(sp0[jp].y, lo1, hi1), sgx_clipf(sp1[jp].y, lo1, hi1));
            }
            if (G::ODD_IN) {
                float s0 = bias0, s1 = bias1;
#pragma unroll
                for (int k = 0; k < CIN / 2; k++) { s0 = fmaf(wk0[k].x, x1[2 * k], s0); s1 = fmaf(wk1[k].x, x1[2 * k], s1); s0 = fmaf(wk0[k].y, x1[2 * k + 1], s0); s1 = fmaf(wk1[k].y, x1[2 * k + 1], s1); }
                const int e = eidx[G::SLOTS_IN - 1];
                if (e >= 0) Em[e & 0xFFFFFF] = (e & (1 << 30)) ? sgx_mk2(0.f, 0.f) : sgx_mk2(sgx_clipf(s0, lo1, hi1), sgx_clipf(s1, lo1, hi1));
            }
        }
        SGX_THREADS_END
        } else {                                                     // one channel per step (scalar 4-byte stores): better where the register budget is tight
        SGX_THREADS_BEGIN(tid)
        SGX_PRIV_BIND(x2, tid); SGX_PRIV_BIND(x1, tid); SGX_PRIV_BIND(eidx, tid);
        float *Ef = (float *)Es;
#pragma unroll UNR_A
        for (int m = 0; m < CM; m++) {
            const sgx_f2 *wr = (const sgx_f2 *)(w1 + (size_t)(cm0 + m) * CIN);
            const float bias = b1[cm0 + m];
            sgx_f2 wk[CIN / 2];
#pragma unroll
            for (int k = 0; k < CIN / 2; k++) wk[k] = wr[k];
            float *Em = Ef + (size_t)(m >> 1) * G::ES * 2 + (m & 1);
            sgx_f2 sp[G::PAIRS_IN ? G::PAIRS_IN : 1];
#pragma unroll
            for (int jp = 0; jp < G::PAIRS_IN; jp++) sp[jp] = sgx_mk2(bias, bias);
#pragma unroll
            for (int k = 0; k < CIN / 2; k++) {                      // the pixel pairs' chains interleaved: independent packed FMAs back to back
#pragma unroll
                for (int jp = 0; jp < G::PAIRS_IN; jp++) sp[jp] = sgx_fma2_wlo(wk[k], x2[jp * CIN + 2 * k], sp[jp]);
#pragma unroll
                for (int jp = 0; jp < G::PAIRS_IN; jp++) sp[jp] = sgx_fma2_whi(wk[k], x2[jp * CIN + 2 * k + 1], sp[jp]);
            }
#pragma unroll
            for (int jp = 0; jp < G::PAIRS_IN; jp++) {
                const sgx_f2 s = sp[jp];
                const int e0 = eidx[2 * jp], e1 = eidx[2 * jp + 1];
                if (e0 >= 0) Em[(e0 & 0xFFFFFF) * 2] = (e0 & (1 << 30)) ? 0.f : sgx_clipf(s.x, lo1, hi1);
                if (e1 >= 0) Em[(e1 & 0xFFFFFF) * 2] = (e1 & (1 << 30)) ? 0.f : sgx_clipf(s.y, lo1, hi1);
            }
            if (G::ODD_IN) {
                float s = bias;
#pragma unroll
                for (int k = 0; k < CIN / 2; k++) { s = fmaf(wk[k].x, x1[2 * k], s); s = fmaf(wk[k].y, x1[2 * k + 1], s); }
                const int e = eidx[G::SLOTS_IN - 1];
                if (e >= 0) Em[(e & 0xFFFFFF) * 2] = (e & (1 << 30)) ? 0.f : sgx_clipf(s, lo1, hi1);
            }
        }
        SGX_THREADS_END
        }
        SGX_SYNC();
        // ---- phases B + C: depthwise on two channels of the chunk at a time, then the project accumulation; one output pixel per thread and slot
        if (PIPE) {
        SGX_THREADS_BEGIN(tid)
        SGX_PRIV_BIND(acc, tid);
        sgx_f2 wk[K * K], bias;
        {
            const sgx_f2 *wt = (const sgx_f2 *)(wd2 + (size_t)cm0 * (K * K));
#pragma unroll
            for (int t = 0; t < K * K; t++) wk[t] = wt[t];
            bias = sgx_mk2(bd[cm0], bd[cm0 + 1]);
        }
#pragma unroll 1
        for (int m = 0; m < CM; m += 2) {
            sgx_f2 wc0[COUT / 2], wc1[COUT / 2], d[G::SLOTS_OUT];
            {   // region 1: fetch the project weights of channel m, run the depthwise taps of the pair
                const sgx_f2 *wp0 = (const sgx_f2 *)(w2t + (size_t)(cm0 + m) * ldw2);
#pragma unroll
                for (int cp = 0; cp < COUT / 2; cp++) wc0[cp] = wp0[cp];
            }
            SGX_SCHED_FENCE();
#pragma unroll
            for (int jo = 0; jo < G::SLOTS_OUT; jo++) {
                const int o = min(tid + SGX_FB2_THREADS * jo, G::NPO - 1), oy = o / TOW, ox = o - oy * TOW;
                const sgx_f2 *e = Es + (size_t)(m >> 1) * G::ES + (oy * S) * G::TIWP + (S == 2 ? ox : ox * S);
                sgx_f2 sv = bias;
#pragma unroll
                for (int a = 0; a < K; a++) {
#pragma unroll
                    for (int c = 0; c < K; c++) sv = sgx_fma2_w(wk[a * K + c], e[a * G::TIWP + (S == 2 ? (c & 1) * G::HALF + (c >> 1) : c)], sv);
                }
                d[jo] = sgx_mk2(sgx_clipf(sv.x, lo2, hi2), sgx_clipf(sv.y, lo2, hi2));
            }
            SGX_SCHED_FENCE();
            {   // region 2: fetch the project weights of channel m + 1, accumulate channel m
                const sgx_f2 *wp1 = (const sgx_f2 *)(w2t + (size_t)(cm0 + m + 1) * ldw2);
#pragma unroll
                for (int cp = 0; cp < COUT / 2; cp++) wc1[cp] = wp1[cp];
            }
            SGX_SCHED_FENCE();
#pragma unroll
            for (int jo = 0; jo < G::SLOTS_OUT; jo++) {
#pragma unroll
                for (int cp = 0; cp < COUT / 2; cp++) acc[jo * (COUT / 2) + cp] = sgx_fma2_w_dlo(wc0[cp], d[jo], acc[jo * (COUT / 2) + cp]);
            }
            SGX_SCHED_FENCE();
            {   // region 3: fetch the depthwise weights of the next pair (the last step reloads its own), accumulate channel m + 1
                const int mn = cm0 + (m + 2 < CM ? m + 2 : m);
                const sgx_f2 *wt = (const sgx_f2 *)(wd2 + (size_t)mn * (K * K));
#pragma unroll
                for (int t = 0; t < K * K; t++) wk[t] = wt[t];
                bias = sgx_mk2(bd[mn], bd[mn + 1]);
            }
            SGX_SCHED_FENCE();
#pragma unroll
            for (int jo = 0; jo < G::SLOTS_OUT; jo++) {
#pragma unroll
                for (int cp = 0; cp < COUT / 2; cp++) acc[jo * (COUT / 2) + cp] = sgx_fma2_w_dhi(wc1[cp], d[jo], acc[jo * (COUT / 2) + cp]);
            }
            SGX_SCHED_FENCE();
        }
        SGX_THREADS_END
        } else {
        SGX_THREADS_BEGIN(tid)
        SGX_PRIV_BIND(acc, tid);
#pragma unroll 1
        for (int m = 0; m < CM; m += 2) {
            const sgx_f2 *wt = (const sgx_f2 *)(wd2 + (size_t)(cm0 + m) * (K * K));                  // pair (cm0 + m) / 2: K * K interleaved weight pairs
            const sgx_f2 *wp0 = (const sgx_f2 *)(w2t + (size_t)(cm0 + m) * ldw2), *wp1 = (const sgx_f2 *)(w2t + (size_t)(cm0 + m + 1) * ldw2);
            const sgx_f2 bias = sgx_mk2(bd[cm0 + m], bd[cm0 + m + 1]);
            sgx_f2 wk[K * K], wc0[COUT / 2], wc1[COUT / 2];
#pragma unroll
            for (int t = 0; t < K * K; t++) wk[t] = wt[t];
#pragma unroll
            for (int cp = 0; cp < COUT / 2; cp++) { wc0[cp] = wp0[cp]; wc1[cp] = wp1[cp]; }
#pragma unroll
            for (int jo = 0; jo < G::SLOTS_OUT; jo++) {
                const int o = min(tid + SGX_FB2_THREADS * jo, G::NPO - 1), oy = o / TOW, ox = o - oy * TOW;
                const sgx_f2 *e = Es + (size_t)(m >> 1) * G::ES + (oy * S) * G::TIWP + (S == 2 ? ox : ox * S);
                sgx_f2 s = bias;
#pragma unroll
                for (int a = 0; a < K; a++) {
#pragma unroll
                    for (int c = 0; c < K; c++) s = sgx_fma2_w(wk[a * K + c], e[a * G::TIWP + (S == 2 ? (c & 1) * G::HALF + (c >> 1) : c)], s);
                }
                const sgx_f2 d = sgx_mk2(sgx_clipf(s.x, lo2, hi2), sgx_clipf(s.y, lo2, hi2));
#pragma unroll
                for (int cp = 0; cp < COUT / 2; cp++) acc[jo * (COUT / 2) + cp] = sgx_fma2_w_dlo(wc0[cp], d, acc[jo * (COUT / 2) + cp]);
#pragma unroll
                for (int cp = 0; cp < COUT / 2; cp++) acc[jo * (COUT / 2) + cp] = sgx_fma2_w_dhi(wc1[cp], d, acc[jo * (COUT / 2) + cp]);
            }
        }
        SGX_THREADS_END
        }
        SGX_SYNC();
    }

    // ---- store (+ residual): all residual loads first, then the adds and the stores
    SGX_THREADS_BEGIN(tid)
    SGX_PRIV_BIND(acc, tid); SGX_PRIV_BIND(rsd, tid);
    const size_t cs = (size_t)Ho * Wo;
#pragma unroll
    for (int jo = 0; jo < G::SLOTS_OUT; jo++) {
        const int o = tid + SGX_FB2_THREADS * jo, oy = o / TOW, ox = o - oy * TOW, gy = oy0 + oy, gx = ox0 + ox;
        if (NQ > 0) {                                                // squeeze-excite gate on the thread's pixel (all threads: no divergence around the scalar weight loads)
            static_assert((NQ & 1) == 0, "squeeze channels are processed in pairs");
            const sgx_f2 *wq1 = (const sgx_f2 *)se.wq1p, *wq2 = (const sgx_f2 *)se.wq2p;
            sgx_f2 hid[NQ / 2 ? NQ / 2 : 1];
#pragma unroll
            for (int jp = 0; jp < NQ / 2; jp++) hid[jp] = sgx_mk2(se.bq1[2 * jp], se.bq1[2 * jp + 1]);
#pragma unroll
            for (int cp = 0; cp < COUT / 2; cp++) {                  // k = 2 cp, 2 cp + 1 ascending; two hidden channels per packed FMA
#pragma unroll
                for (int jp = 0; jp < NQ / 2; jp++) hid[jp] = sgx_fma2_w_dlo(wq1[jp * COUT + 2 * cp], acc[jo * (COUT / 2) + cp], hid[jp]);
#pragma unroll
                for (int jp = 0; jp < NQ / 2; jp++) hid[jp] = sgx_fma2_w_dhi(wq1[jp * COUT + 2 * cp + 1], acc[jo * (COUT / 2) + cp], hid[jp]);
            }
#pragma unroll
            for (int jp = 0; jp < NQ / 2; jp++) hid[jp] = sgx_mk2(sgx_clipf(hid[jp].x, se.qlo, se.qhi), sgx_clipf(hid[jp].y, se.qlo, se.qhi));
            sgx_f2 gt[COUT / 2];
#pragma unroll
            for (int cp = 0; cp < COUT / 2; cp++) gt[cp] = sgx_mk2(se.bq2[2 * cp], se.bq2[2 * cp + 1]);
#pragma unroll
            for (int j = 0; j < NQ; j++) {
#pragma unroll
                for (int cp = 0; cp < COUT / 2; cp++) gt[cp] = (j & 1) ? sgx_fma2_w_dhi(wq2[j * (COUT / 2) + cp], hid[j >> 1], gt[cp]) : sgx_fma2_w_dlo(wq2[j * (COUT / 2) + cp], hid[j >> 1], gt[cp]);
            }
#pragma unroll
            for (int cp = 0; cp < COUT / 2; cp++) {                  // [ADD c][CLIP][DIV c][MUL project output], as sgx_epi_mode<SGX_EMODE_GATE>
                float ux = gt[cp].x + se.gc1, uy = gt[cp].y + se.gc1;
                ux = sgx_clipf(ux, se.glo, se.ghi); uy = sgx_clipf(uy, se.glo, se.ghi);
                ux = ux / se.gc2; uy = uy / se.gc2;
                acc[jo * (COUT / 2) + cp] = sgx_mk2(ux * acc[jo * (COUT / 2) + cp].x, uy * acc[jo * (COUT / 2) + cp].y);
            }
        }
        if (o < G::NPO && gy < Ho && gx < Wo) {
            const size_t off0 = (size_t)gy * Wo + gx;
            float v[COUT];
#pragma unroll
            for (int co = 0; co < COUT; co++) v[co] = (co & 1) ? acc[jo * (COUT / 2) + (co >> 1)].y : acc[jo * (COUT / 2) + (co >> 1)].x;
            if (RES == 1) {
#pragma unroll
                for (int co = 0; co < COUT; co++) v[co] = v[co] + rsd[jo * COUT + co];
            } else if (RES == 2) {                                   // all loads first, then the adds
                const float *rp = res + (size_t)b * res_pitch + off0;
                float r[COUT];
#pragma unroll
                for (int co = 0; co < COUT; co++) r[co] = rp[(size_t)co * cs];
#pragma unroll
                for (int co = 0; co < COUT; co++) v[co] = v[co] + r[co];
            }
            float *op = out + (size_t)b * out_pitch + off0;
#pragma unroll
            for (int co = 0; co < COUT; co++) op[(size_t)co * cs] = v[co];
        }
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_se_gate<COUT, NQ>: a squeeze-excite tail on its own — out = clip(bq2 + Wq2 x clip(bq1 + Wq1 x y, qlo, qhi) + gc1, glo, ghi) / gc2 * y [+ residual] for every pixel of y.
// The per-layer plan runs it as two pointwise launches (COUT -> NQ with ReLU, NQ -> COUT with the gate epilogue): with COUT = 40 and NQ = 10 those are 400 + 400 multiply-adds
// per pixel behind 32-wide matrix-core tiles (a quarter of the rows used) and three passes over y.  Here a thread owns a pixel: COUT loads (coalesced across the wave), the two
// matrix-vector products on wave-uniform weights exactly as in k_fused_block2's tail (same fmaf chains, k ascending from the bias), COUT stores.  grid = ceil(B * HW / 256).
// ---------------------------------------------------------------------------------------------
struct SgxSeGate { int Cout, Cq, HW; const float *y; size_t y_pitch; float *out; size_t out_pitch; const float *res; size_t res_pitch; SgxFb2Se se; const float *wq1, *wq2; };
template <int COUT, int NQ>
SGX_KERNEL(256) k_se_gate(int HW, int total, const float *__restrict__ y, size_t y_pitch, float *__restrict__ out, size_t out_pitch, const float *__restrict__ res, size_t res_pitch, SgxFb2Se se)
{
    static_assert((NQ & 1) == 0 && (COUT & 1) == 0, "channels are processed in pairs");
    SGX_THREADS_BEGIN(tid)
    const int g = min((int)blockIdx.x * 256 + tid, total - 1);            // the lanes past the end recompute the last pixel (no divergence around the scalar weight loads) and do not store
    const bool live = (int)blockIdx.x * 256 + tid < total;
    const int b = g / HW, n = g - b * HW;
    const float *yp = y + (size_t)b * y_pitch + n;
    sgx_f2 acc[COUT / 2];
#pragma unroll
    for (int cp = 0; cp < COUT / 2; cp++) acc[cp] = sgx_mk2(yp[(size_t)(2 * cp) * HW], yp[(size_t)(2 * cp + 1) * HW]);
    float r[COUT];
    if (res) {
        const float *rp = res + (size_t)b * res_pitch + n;
#pragma unroll
        for (int co = 0; co < COUT; co++) r[co] = rp[(size_t)co * HW];
    }
    const sgx_f2 *wq1 = (const sgx_f2 *)se.wq1p, *wq2 = (const sgx_f2 *)se.wq2p;
    sgx_f2 hid[NQ / 2];
#pragma unroll
    for (int jp = 0; jp < NQ / 2; jp++) hid[jp] = sgx_mk2(se.bq1[2 * jp], se.bq1[2 * jp + 1]);
#pragma unroll
    for (int cp = 0; cp < COUT / 2; cp++) {
#pragma unroll
        for (int jp = 0; jp < NQ / 2; jp++) hid[jp] = sgx_fma2_w_dlo(wq1[jp * COUT + 2 * cp], acc[cp], hid[jp]);
#pragma unroll
        for (int jp = 0; jp < NQ / 2; jp++) hid[jp] = sgx_fma2_w_dhi(wq1[jp * COUT + 2 * cp + 1], acc[cp], hid[jp]);
    }
#pragma unroll
    for (int jp = 0; jp < NQ / 2; jp++) hid[jp] = sgx_mk2(sgx_clipf(hid[jp].x, se.qlo, se.qhi), sgx_clipf(hid[jp].y, se.qlo, se.qhi));
    sgx_f2 gt[COUT / 2];
#pragma unroll
    for (int cp = 0; cp < COUT / 2; cp++) gt[cp] = sgx_mk2(se.bq2[2 * cp], se.bq2[2 * cp + 1]);
#pragma unroll
    for (int j = 0; j < NQ; j++) {
#pragma unroll
        for (int cp = 0; cp < COUT / 2; cp++) gt[cp] = (j & 1) ? sgx_fma2_w_dhi(wq2[j * (COUT / 2) + cp], hid[j >> 1], gt[cp]) : sgx_fma2_w_dlo(wq2[j * (COUT / 2) + cp], hid[j >> 1], gt[cp]);
    }
    float *op = out + (size_t)b * out_pitch + n;
#pragma unroll
    for (int cp = 0; cp < COUT / 2; cp++) {                                // [ADD c][CLIP][DIV c][MUL y] [ADD residual], as sgx_epi_mode<SGX_EMODE_GATE / GATE_ADD>
        float ux = gt[cp].x + se.gc1, uy = gt[cp].y + se.gc1;
        ux = sgx_clipf(ux, se.glo, se.ghi); uy = sgx_clipf(uy, se.glo, se.ghi);
        ux = ux / se.gc2; uy = uy / se.gc2;
        ux = ux * acc[cp].x; uy = uy * acc[cp].y;
        if (res) { ux = ux + r[2 * cp]; uy = uy + r[2 * cp + 1]; }
        if (live) { op[(size_t)(2 * cp) * HW] = ux; op[(size_t)(2 * cp + 1) * HW] = uy; }
    }
    SGX_THREADS_END
}
static inline bool sgx_se_gate_supported(int cout, int cq) { return cout == 40 && cq == 10; }
static inline int sgx_se_gate_launch(const SgxSeGate &p, int batch, sgx_stream_t st)
{
    const int total = batch * p.HW;
    if (p.Cout == 40 && p.Cq == 10) { auto kfn = k_se_gate<40, 10>; SGX_LAUNCH(kfn, dim3((unsigned)((total + 255) / 256)), dim3(256), st, p.HW, total, p.y, p.y_pitch, p.out, p.out_pitch, p.res, p.res_pitch, p.se); return SGX_OK; }
    return SGX_ERR_INVALID;
}

#ifdef SGX_DEBUG_TAPS
// ---------------------------------------------------------------------------------------------
// k_conv_dw3<K, S, PX>: depthwise K x K convolution (+ ReLU / Clip / h-swish epilogue) in the arithmetic of k_fused_block2 / k_hrb (round 6).
// k_conv_dw2 spends 2.7 lane-instructions per multiply-add (round-6 census: the 100 fp32 FMAs of a four-pixel task are a quarter of its cycles; the rest is the weight vector
// read from LDS per task, index arithmetic, a scalar epilogue per pixel).  Here the CHANNEL PAIR is the unit:
//   stage    the input planes of NP channel pairs of one image -> LDS, pair-interleaved [pair][padded row][padded column] float2 with the zero border in place
//            (two coalesced global loads + one ds_write_b64 per pixel and pair)
//   compute  wave w takes pairs w, w + 4, ...; the K x K weight pairs of a pair sit in SCALAR registers (wave-uniform), a lane takes PX horizontally adjacent output pixels
//            (S = 1: their tap windows overlap, (PX + K - 1) x K ds_read_b64 serve PX K K v_pk_fma_f32), taps (i, j) ascending from the bias — per output the same fmaf chain as
//            k_conv_dw2 / k_conv_kxk, so the results are BIT-IDENTICAL to them — and the epilogue program runs on both halves of the pair
// grid = images x ceil(pairs / NP); LDS = NP * HP * WP * 8 bytes.
// MEASURED SLOWER than k_conv_dw2 (0.259 / 0.254 / 0.266 / 0.222 ms against 0.181 / 0.181 / 0.253 / 0.181 on the four big depthwise steps, 512 frames): 15 KB of LDS per channel
// pair leave three pairs per workgroup (one wave idle, three workgroups per CU) and the stage / compute phases do not overlap.  Tap build only (SGX_DW3=1).
// ---------------------------------------------------------------------------------------------
struct SgxDw3 { int C, H, W, Ho, Wo, pad, NP, HP, WP, nchunks, WU, NU; unsigned m_wu, m_wp, m_cells; const float *in; size_t in_pitch; const float *wd2, *bias; float *out; size_t out_pitch; SgxEpi epi; };
template <int K, int S, int PX, int MODE>
SGX_DEV void sgx_dw3_body(const SgxDw3 &p, sgx_f2 *Es, int tid)
{
    constexpr int KK = K * K, NC = (PX - 1) * S + K;                     // tap columns a lane reads per tap row
    const int chunk = (int)blockIdx.x % p.nchunks, b = (int)blockIdx.x / p.nchunks;
    const int pair0 = chunk * p.NP, np = min(p.NP, p.C / 2 - pair0);
    const int wave = tid >> 6, lane = tid & 63;
    const size_t ohw = (size_t)p.Ho * p.Wo;
    float *Y = p.out + (size_t)b * p.out_pitch;
    for (int pl = wave; pl < np; pl += 4) {                              // wave-uniform
        const int gp = pair0 + pl;
        const sgx_f2 *wt = (const sgx_f2 *)p.wd2 + (size_t)gp * KK;
        sgx_f2 wk[KK];
#pragma unroll
        for (int t = 0; t < KK; t++) wk[t] = wt[t];
        const sgx_f2 bz = sgx_mk2(p.bias[2 * gp], p.bias[2 * gp + 1]);
        const sgx_f2 *E = Es + (size_t)pl * p.HP * p.WP;
        for (int u0 = 0; u0 < p.NU; u0 += 64) {
            const int u = min(u0 + lane, p.NU - 1), oy = (int)sgx_fastdiv((unsigned)u, p.m_wu), oxu = u - oy * p.WU, ox = oxu * PX;      // NU < 2^20, WU < 2^12
            const sgx_f2 *e = E + (oy * S) * p.WP + ox * S;
            sgx_f2 sv[PX];
#pragma unroll
            for (int i = 0; i < PX; i++) sv[i] = bz;
#pragma unroll
            for (int a = 0; a < K; a++) {
                sgx_f2 tp[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) tp[c] = e[a * p.WP + c];
#pragma unroll
                for (int c = 0; c < K; c++)
#pragma unroll
                    for (int i = 0; i < PX; i++) sv[i] = sgx_fma2_w(wk[a * K + c], tp[c + i * S], sv[i]);
            }
            if (u0 + lane < p.NU) {
#pragma unroll
                for (int i = 0; i < PX; i++)
                    if (ox + i < p.Wo) {
                        const size_t o = (size_t)oy * p.Wo + ox + i;
                        Y[(size_t)(2 * gp) * ohw + o] = sgx_epi_mode<MODE>(p.epi, sv[i].x, 0, 0);
                        Y[(size_t)(2 * gp + 1) * ohw + o] = sgx_epi_mode<MODE>(p.epi, sv[i].y, 0, 0);
                    }
            }
        }
    }
}
template <int K, int S, int PX>
SGX_KERNEL(256) k_conv_dw3(SgxDw3 p)
{
    SGX_DYN_LDS(smem);
    sgx_f2 *Es = (sgx_f2 *)smem;
    const int chunk = (int)blockIdx.x % p.nchunks, b = (int)blockIdx.x / p.nchunks;
    const int pair0 = chunk * p.NP, np = min(p.NP, p.C / 2 - pair0);
    const float *X = p.in + (size_t)b * p.in_pitch;
    const size_t hw = (size_t)p.H * p.W;
    const int cells = p.HP * p.WP, total = np * cells;
    SGX_THREADS_BEGIN(tid)
    for (int i0 = tid; i0 < total; i0 += 256 * 4) {                      // four cells per thread and round: eight independent loads in flight
        sgx_f2 v[4]; int idx[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = min(i0 + 256 * q, total - 1), pl = p.m_cells ? (int)sgx_fastdiv((unsigned)i, p.m_cells) : i / cells, r = i - pl * cells, yy = (int)sgx_fastdiv((unsigned)r, p.m_wp), xx = r - yy * p.WP, iy = yy - p.pad, ix = xx - p.pad;
            const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float *src = X + (size_t)(2 * (pair0 + pl)) * hw + (size_t)(in ? iy : 0) * p.W + (in ? ix : 0);
            const float x0 = src[0], x1 = src[hw];
            v[q] = sgx_mk2(in ? x0 : 0.f, in ? x1 : 0.f); idx[q] = i0 + 256 * q < total ? i : -1;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) if (idx[q] >= 0) Es[idx[q]] = v[q];
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    switch (p.epi.mode) {
    case SGX_EMODE_NONE: sgx_dw3_body<K, S, PX, SGX_EMODE_NONE>(p, Es, tid); break;
    case SGX_EMODE_ACT: sgx_dw3_body<K, S, PX, SGX_EMODE_ACT>(p, Es, tid); break;
    default: sgx_dw3_body<K, S, PX, SGX_EMODE_HSWISH>(p, Es, tid); break;      // the host launches this kernel for these three programs only
    }
    SGX_THREADS_END
}
static inline bool sgx_dw3_plan(int C, int H, int W, int Ho, int Wo, int k, int stride, int pad, int mode, SgxDw3 *p, int *px, size_t *lds)
{
    if ((C & 1) || (k != 3 && k != 5) || (stride != 1 && stride != 2) || pad != k / 2 || (mode != SGX_EMODE_NONE && mode != SGX_EMODE_ACT && mode != SGX_EMODE_HSWISH)) return false;
    const int HP = H + 2 * pad, WP = W + 2 * pad + 3;                   // + 3: the tap window of the last (partial) pixel group stays inside the row
    const size_t per_pair = (size_t)HP * WP * 8;
    if (per_pair > 60 * 1024) return false;
    int NP = (int)((56 * 1024) / per_pair); NP = NP > 16 ? 16 : NP; NP = NP > C / 2 ? C / 2 : NP; NP = NP < 1 ? 1 : NP;
    if (NP >= 4) NP &= ~3;                                               // whole rounds of the four waves
    *px = (stride == 1 && Wo >= 16) ? 2 : 1;
    p->C = C; p->H = H; p->W = W; p->Ho = Ho; p->Wo = Wo; p->pad = pad; p->NP = NP; p->HP = HP; p->WP = WP; p->nchunks = (C / 2 + NP - 1) / NP;
    p->WU = (Wo + *px - 1) / *px; p->NU = Ho * p->WU;
    { auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
      p->m_wu = magic(p->WU); p->m_wp = magic(WP); p->m_cells = (HP * WP < 4096 && NP * HP * WP < (1 << 20)) ? magic(HP * WP) : 0u; }
    *lds = (size_t)NP * per_pair;
    return true;
}
static inline void sgx_dw3_launch(const SgxDw3 &p, int k, int stride, int px, size_t lds, int batch, sgx_stream_t st)
{
    const unsigned grid = (unsigned)(batch * p.nchunks);
#define SGX_DW3(K_, S_, PX_) do { auto kfn = k_conv_dw3<K_, S_, PX_>; SGX_LAUNCH_DYN(kfn, dim3(grid), dim3(256), lds, st, p); } while (0)
    if (k == 3 && stride == 1) { if (px == 2) SGX_DW3(3, 1, 2); else SGX_DW3(3, 1, 1); }
    else if (k == 3) SGX_DW3(3, 2, 1);
    else if (stride == 1) { if (px == 2) SGX_DW3(5, 1, 2); else SGX_DW3(5, 1, 1); }
    else SGX_DW3(5, 2, 1);
#undef SGX_DW3
}

#endif

// ---- k_fused_block2 dispatch: (Cin, Cout, K, stride) -> instantiation; the tile variant comes from SGX_FB2_TILE (tuning tap) --------------------------------------
static inline int sgx_fb2_cm(int v2) { return (v2 >> 4) == 1 ? 16 : 8; }       /* expanded channels per chunk of the instantiation (Cmid must be a multiple) */
static inline int sgx_fb2_variant(int cin, int cout, int k, int stride, int cq = 0)
{
    if (cq) {                                                                   // block with a squeeze-excite tail: 24 -> 72 -> 40 (5 x 5, stride 2, 75 -> 38): 0.86 -> 0.75 ms per 512 frames.
        // The two 40 -> 120 -> 40 blocks at 38 x 38 were measured too (8 x 16 tiles): 0.67 ms against 0.61 for the five per-layer kernels — 40 input channels leave one pixel
        // pair per thread (a single dependent FMA chain in phase A) and the 5 x 5 halo doubles the expand work; they stay on the per-layer plan.
        if (cin == 24 && cout == 40 && k == 5 && stride == 2 && cq == 10) return 4 * 16;
        return 0;
    }
    static const int tile_env = sgx_getenv("SGX_FB2_TILE") ? atoi(sgx_getenv("SGX_FB2_TILE")) : -1;          // tuning tap: force 8 x 16 (0) or 16 x 16 (1) output tiles on the stride-1 blocks
    int shape = 0;
    if (cin == 16 && cout == 16 && k == 3 && stride == 1) shape = 1;
    else if (cin == 16 && cout == 24 && k == 3 && stride == 2) shape = 2;
    else if (cin == 24 && cout == 24 && k == 3 && stride == 1) shape = 3;
    if (!shape) return 0;
    const int tile = tile_env >= 0 ? tile_env : (shape == 1 ? 1 : 0);           // measured at 512 frames: 16 -> 16 -> 16 at 150 x 150 is best with 16 x 16 tiles (0.76 ms against 0.79), 24 -> 72 -> 24 with 8 x 16 (0.61 against 0.68)
    return shape * 16 + ((shape == 2) ? 0 : (tile & 1));                       // stride 1: tile 0 = 8 x 16, tile 1 = 16 x 16; stride 2: 8 x 16 only (input pixels live in registers)
}
static inline void sgx_fb2_tile(int v2, int *toh, int *tow)
{
    *tow = 16;
    if ((v2 >> 4) == 4) { *toh = 4; *tow = 19; return; }        // 38 = 2 x 19 columns: 20 tiles per image; measured 0.67 ms against 0.75 (5 x 16) and 0.72 (6 x 13)
    *toh = (v2 >> 4) == 2 ? 7 : (v2 & 1) ? 16 : 8;
}       // stride 2: 7 x 16 outputs = 15 x 33 inputs = 4 full slots
static inline int sgx_fb2_launch(const SgxFusedBlk &fb, int batch, sgx_stream_t st)
{
    const unsigned grid = (unsigned)(fb.tiles_x * fb.tiles_y * batch);
    SgxFb2Se se; se.wq1p = fb.wq1p; se.bq1 = fb.bq1; se.wq2p = fb.wq2p; se.bq2 = fb.bq2; se.qlo = fb.qlo; se.qhi = fb.qhi; se.gc1 = fb.gc1; se.glo = fb.glo; se.ghi = fb.ghi; se.gc2 = fb.gc2;
#define SGX_FB2Q(CIN_, COUT_, K_, S_, TOH_, TOW_, CM_, RES_, UA_, NQ_, PIPE_) do { auto kfn = fb.res ? k_fused_block2<CIN_, COUT_, K_, S_, TOH_, TOW_, CM_, RES_, UA_, NQ_, PIPE_> : k_fused_block2<CIN_, COUT_, K_, S_, TOH_, TOW_, CM_, 0, UA_, NQ_, PIPE_>;                                        \
        SGX_LAUNCH(kfn, dim3(grid), dim3(SGX_FB2_THREADS), st, fb.Cmid, fb.H, fb.W, fb.Ho, fb.Wo, fb.pad, fb.tiles_x, fb.tiles_y, fb.lo1, fb.hi1, fb.lo2, fb.hi2,     \
                   fb.in, fb.in_pitch, fb.w1, fb.b1, fb.wd2, fb.bd, fb.w2t, fb.ldw2, fb.b2, fb.out, fb.out_pitch, fb.res, fb.res_pitch, se); } while (0)
#define SGX_FB2(CIN_, COUT_, K_, S_, TOH_, TOW_, CM_, RES_, UA_) SGX_FB2Q(CIN_, COUT_, K_, S_, TOH_, TOW_, CM_, RES_, UA_, 0, 0)
    switch (fb.v2) {                                                   // pipelined weight fetch (PIPE) measured on every shape: 24 -> 72 -> 40 gains (0.86 -> 0.74 ms), the three 3 x 3 shapes lose 0.04 - 0.08 ms
    case 16: SGX_FB2(16, 16, 3, 1, 8, 16, 16, 1, 3); break;           // Cmid = 16: one chunk
    case 17: SGX_FB2(16, 16, 3, 1, 16, 16, 16, 1, 3); break;
    case 32: SGX_FB2(16, 24, 3, 2, 7, 16, 8, 2, 2); break;           // phase A on channel pairs: 0.83 -> 0.70 ms (the other two shapes lose: 0.71 -> 0.76, 0.61 -> 0.72)
    case 48: SGX_FB2(24, 24, 3, 1, 8, 16, 8, 2, 1); break;
    case 49: SGX_FB2(24, 24, 3, 1, 16, 16, 8, 2, 1); break;
    case 64: SGX_FB2Q(24, 40, 5, 2, 4, 19, 8, 2, 1, 10, 1); break;     // 4 x 19 outputs = 11 x 41 inputs = 4 slots of 128 threads
    default: return SGX_ERR_INVALID;
    }
#undef SGX_FB2
#undef SGX_FB2Q
    return SGX_OK;
}
