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
};
#define SGX_FB_CM 32                                           /* expanded channels per chunk = one MFMA row block */
static inline size_t sgx_fb_lds_floats(const SgxFusedBlk &p)
{
    const int coP = ((p.Cout + 31) / 32) * 32;
    return (size_t)p.Cin * p.NPI + (size_t)p.Cin * SGX_FB_CM + (size_t)p.CMR * p.ES + (size_t)SGX_FB_CM * p.K * p.K + 2 * SGX_FB_CM + (size_t)p.CMR * p.NPO +
           (size_t)p.CMR * coP;
}

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
