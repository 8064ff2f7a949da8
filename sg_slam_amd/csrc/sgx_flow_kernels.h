// sgx_flow_kernels.h — the inputs of the dynamic-feature mask on the device (tier N1 of SURVEY.md §8):
//   cv::calcOpticalFlowPyrLK(imGray, imGrayPre, Curpoint, Prepoint, State, Err, Size(21,21), 3, TermCriteria(ITER|EPS, 30, 0.01))
//                                                                                   src/sg-slam/src/Frame.cc:445
//   point selection outside the previous frame's person boxes                       src/sg-slam/src/Frame.cc:454-467
//   cv::findFundamentalMat(cur, prev, FM_RANSAC, 1.0, 0.99)                         src/sg-slam/src/Frame.cc:469-472
// OpenCV 3.4 algorithms as restated in oracle/flow_oracle.c (lkpyramid.cpp, pyramids.cpp, fundam.cpp, ptsetreg.cpp).
//
// Kernels
//   k_lk_copy       level 0 of the LK pyramid = a pitch-aligned copy of the frame (it becomes "imGrayPre" of the next call)
//   k_lk_pyrdown    cv::pyrDown: 5x5 [1 4 6 4 1]^2, (sum + 128) >> 8, REFLECT_101 — four outputs per thread from aligned dwords
//   k_lk_track      LKTrackerInvoker: ONE WAVE PER KEYPOINT, all pyramid levels chained in one launch.  Lane (r, s) owns window row r and the seven
//                   columns 7s..7s+6 (63 lanes = 21 x 21 samples): its window intensities and Scharr derivatives (calcSharrDeriv evaluated on the fly from an
//                   LDS-staged patch of the image — no derivative plane ever exists in HBM) stay in registers for all iterations; the
//                   tracked-to image patch sits in a per-wave LDS tile (32 rows x 36 B, re-staged only when the window walks out of it).  Every sum
//                   (2x2 gradient matrix, mismatch vector) is accumulated as EXACT integers — per lane in 32 bits, across the wave as two 16-bit
//                   halves through DPP row reductions — and converted to float once: the order-free variant of OpenCV's accumulation
//                   (`typedef int64 acctype`), bit-identical to oracle acc_mode 1.  The per-iteration float arithmetic follows lkpyramid.cpp operation
//                   by operation (-ffp-contract=off).
//   k_fm_ransac     one workgroup per frame: pair selection (block scan), then RANSAC (LMedS for 8 .. 14 pairs, as OpenCV) in chunks: thread 0 advances cv::RNG and forms the 7-index
//                   groups, a thread per group runs checkSubset + run7Point (Householder null space, cv::solveCubic) in fp64, all threads score the
//                   models (FMEstimatorCallback::computeError), thread 0 replays the sequential accept / RANSACUpdateNumIters rule in iteration
//                   order — same samples, same winner as the sequential library loop.
#pragma once
#include "sgx_rt.h"
#include "sgx_block.h"
#include <float.h>
#include <math.h>

#define SGX_LK_MAXL 4                 /* pyramid levels (maxLevel <= 3, Frame.cc:445 passes 3) */
#define SGX_LK_WIN 21                 /* winSize (the lane mapping of k_lk_track is built for 21 x 21) */
#define SGX_LK_TILE_PITCH 36          /* bytes per LDS tile row: 9 dwords (odd dword stride) */
struct SgxLkGeom {
    int nl;
    int w[SGX_LK_MAXL], h[SGX_LK_MAXL], pitch[SGX_LK_MAXL];
    unsigned ioff[SGX_LK_MAXL];       /* byte offset of level l inside a frame's image block */
    unsigned img_stride;              /* bytes per frame */
};

SGX_DEV int sgx_reflect101(int p, int len)      /* cv::borderInterpolate(p, len, BORDER_REFLECT_101), any p */
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}
SGX_DEV int sgx_floor_f(float v) { return (int)floorf(v); }
/* exact int64 -> float (round to nearest even), |v| < 2^47: (v >> 24) and the 24-bit remainder are both exact floats, one rounding in the add */
SGX_DEV float sgx_i64_to_f32(long long v)
{
    const long long hi = v >> 24; const int lo = (int)(v - (hi << 24));
    return (float)(int)hi * 16777216.0f + (float)lo;
}

// ---------------------------------------------------------------------------------------------
// exact n / d for n < 2^21, d < 2^11 with m = ceil(2^32 / d) from the host (0 encodes d == 1): one mul-hi instead of the ~25-instruction division sequence — which was most of k_lk_copy
SGX_DEV unsigned sgx_lk_udiv(unsigned n, unsigned m)
{
#ifndef SGX_EMU
    return m ? __umulhi(n, m) : n;
#else
    return m ? (unsigned)(((unsigned long long)n * m) >> 32) : n;
#endif
}
SGX_KERNEL(256) k_lk_copy(const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch, unsigned dstride, unsigned qw_magic)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, qw = dpitch >> 2, q = (int)blockIdx.x * 256 + tid;
    if (q < qw * h) {
        const int y = (int)sgx_lk_udiv((unsigned)q, qw_magic), x = (q - y * qw) * 4;
        const uint8_t *s = src + ((size_t)f * h + y) * spitch;
        uint32_t v;
        if (x + 4 <= w) v = *(const uint32_t *)(s + x);
        else { v = 0; for (int i = 0; x + i < w; i++) v |= (uint32_t)s[x + i] << (8 * i); }
        *(uint32_t *)(dst + (size_t)f * dstride + (size_t)y * dpitch + x) = v;
    }
    SGX_THREADS_END
}

// cv::pyrDown (pyramids.cpp, pyrDown_<FixPtCast<uchar,8>>): dst(x, y) = (sum_{i,j} k_i k_j src(2x-2+i, 2y-2+j) + 128) >> 8, k = [1 4 6 4 1]
SGX_KERNEL(256) k_lk_pyrdown(const uint8_t *src, int sw, int sh, int spitch, unsigned sstride, uint8_t *dst, int dw, int dh, int dpitch, unsigned dstride, unsigned qw_magic)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, qw = dpitch >> 2, q = (int)blockIdx.x * 256 + tid;
    if (q < qw * dh) {
        const int y = (int)sgx_lk_udiv((unsigned)q, qw_magic), X = (q - y * qw) * 4;
        const uint8_t *S = src + (size_t)f * sstride;
        int acc[4] = { 0, 0, 0, 0 };
        const bool fast = X >= 2 && 2 * X + 11 < spitch && 2 * X + 8 <= sw - 1;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int wk = k == 0 || k == 4 ? 1 : k == 2 ? 6 : 4;
            const uint8_t *row = S + (size_t)sgx_reflect101(2 * y - 2 + k, sh) * spitch;
            int p[11];
            if (fast) {
                const uint32_t *r4 = (const uint32_t *)(row + 2 * X - 4);
                const uint32_t a = r4[0], b = r4[1], c = r4[2], d = r4[3];
                p[0] = (a >> 16) & 255; p[1] = a >> 24;
                p[2] = b & 255; p[3] = (b >> 8) & 255; p[4] = (b >> 16) & 255; p[5] = b >> 24;
                p[6] = c & 255; p[7] = (c >> 8) & 255; p[8] = (c >> 16) & 255; p[9] = c >> 24;
                p[10] = d & 255;
            } else {
#pragma unroll
                for (int i = 0; i < 11; i++) p[i] = row[sgx_reflect101(2 * X - 2 + i, sw)];
            }
#pragma unroll
            for (int o = 0; o < 4; o++) acc[o] += wk * (p[2 * o + 2] * 6 + (p[2 * o + 1] + p[2 * o + 3]) * 4 + p[2 * o] + p[2 * o + 4]);
        }
        uint32_t out = 0;
#pragma unroll
        for (int o = 0; o < 4; o++) out |= (uint32_t)((acc[o] + 128) >> 8) << (8 * o);       /* columns >= dw of the last quad are padding */
        *(uint32_t *)(dst + (size_t)f * dstride + (size_t)y * dpitch + X) = out;
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// LKTrackerInvoker
// ---------------------------------------------------------------------------------------------
struct SgxLkWeights { int w00, w01, w10, w11; };
SGX_DEV SgxLkWeights sgx_lk_weights(float a, float b)            /* lkpyramid.cpp: W_BITS = 14, iw11 is the remainder */
{
    SgxLkWeights k;
    k.w00 = sgx_cvround((1.f - a) * (1.f - b) * (float)(1 << 14));
    k.w01 = sgx_cvround(a * (1.f - b) * (float)(1 << 14));
    k.w10 = sgx_cvround((1.f - a) * b * (float)(1 << 14));
    k.w11 = (1 << 14) - k.w00 - k.w01 - k.w10;
    return k;
}
#define SGX_LK_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* the part of one iteration that is the same on every lane: OpenCV's float arithmetic, operation by operation */
struct SgxLkStep { float dx, dy; };
SGX_DEV SgxLkStep sgx_lk_step(float A11, float A12, float A22, float D, long long sb1, long long sb2)
{
    const float FLT_SCALE = 1.f / (1 << 20);
    const float b1 = sgx_i64_to_f32(sb1) * FLT_SCALE, b2 = sgx_i64_to_f32(sb2) * FLT_SCALE;
    SgxLkStep s;
    s.dx = (A12 * b2 - A22 * b1) * D;
    s.dy = (A12 * b1 - A11 * b2) * D;
    return s;
}

struct SgxLkArgs {
    const uint8_t *cur_img, *prev_img;       /* per-frame image blocks (SgxLkGeom offsets) */
    const uint8_t *keys; const int *n; int cap;
    float *prev_xy; uint8_t *status;
    int max_count; double eps2; float min_eig;
    int batch, kblocks;                      /* frames, keypoint blocks (4 keypoints each) per frame */
};
/* block -> (frame, keypoint block): blocks are dealt to the 8 XCDs round-robin (observed dispatch order, speed only), so block i works for frame
 * (i % 8) + 8 * slot and walks through ALL keypoint blocks of that frame before the XCD moves to its next frame: the two pyramids of a frame
 * (0.8 MB) stay in one XCD's 4 MB L2 while its ~1000 windows are gathered from them. */
SGX_DEV void sgx_lk_decode_block(int i, int batch, int kblocks, int &f, int &kb)
{
    const int full = (batch / 8) * 8;                              /* frames that fill whole groups of eight */
    if (i < full * kblocks) { const int x = i & 7, j = i >> 3; kb = j % kblocks; f = x + 8 * (j / kblocks); }
    else { const int j = i - full * kblocks; f = full + j / kblocks; kb = j % kblocks; }
}

#ifdef SGX_EMU
// scalar twin of the wave kernel (kernel-logic emulator only): same integer sums, same float steps
static inline int sgx_lk_px(const uint8_t *I, int w, int h, int pitch, int x, int y) { return I[(size_t)sgx_reflect101(y, h) * pitch + sgx_reflect101(x, w)]; }
static void sgx_lk_track_point(const SgxLkGeom &g, const uint8_t *I0, const uint8_t *J0, float kx, float ky, int max_count, double eps2, float min_eig,
                               float *ox, float *oy, uint8_t *ost)
{
    const int W = SGX_LK_WIN; const float half = (W - 1) * 0.5f, FLT_SCALE = 1.f / (1 << 20);
    float outx = 0.f, outy = 0.f; uint8_t status = 1;
    static thread_local short Iw[SGX_LK_WIN * SGX_LK_WIN], dIw[SGX_LK_WIN * SGX_LK_WIN * 2];
    for (int level = g.nl - 1; level >= 0; level--) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        const uint8_t *I = I0 + g.ioff[level], *J = J0 + g.ioff[level];
        const float sc = 1.0f / (float)(1 << level);
        float prevx = kx * sc, prevy = ky * sc, nextx, nexty;
        if (level == g.nl - 1) { nextx = prevx; nexty = prevy; } else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;
        prevx -= half; prevy -= half;
        const int ipx = sgx_floor_f(prevx), ipy = sgx_floor_f(prevy);
        if (ipx < -W || ipx >= w || ipy < -W || ipy >= h) { if (level == 0) status = 0; continue; }
        SgxLkWeights k = sgx_lk_weights(prevx - ipx, prevy - ipy);
        long long s11 = 0, s12 = 0, s22 = 0;
        for (int y = 0; y < W; y++) for (int x = 0; x < W; x++) {
            int pv[4], dx[4], dy[4];
            for (int c = 0; c < 4; c++) {
                const int gx = ipx + x + (c & 1), gy = ipy + y + (c >> 1);
                pv[c] = sgx_lk_px(I, w, h, pitch, gx, gy);
                if (gx < 0 || gy < 0 || gx >= w || gy >= h) { dx[c] = dy[c] = 0; }
                else {          /* calcSharrDeriv on the REFLECT_101-extended image; outside the image the derivative plane is 0 (copyMakeBorder CONSTANT) */
                    int t0[3], t1[3];
                    for (int q = 0; q < 3; q++) {
                        const int a = sgx_lk_px(I, w, h, pitch, gx - 1 + q, gy - 1), b = sgx_lk_px(I, w, h, pitch, gx - 1 + q, gy), cc = sgx_lk_px(I, w, h, pitch, gx - 1 + q, gy + 1);
                        t0[q] = (a + cc) * 3 + b * 10; t1[q] = cc - a;
                    }
                    dx[c] = t0[2] - t0[0]; dy[c] = (t1[0] + t1[2]) * 3 + t1[1] * 10;
                }
            }
            const int iv = SGX_LK_DESCALE(pv[0] * k.w00 + pv[1] * k.w01 + pv[2] * k.w10 + pv[3] * k.w11, 9);
            const int ix = SGX_LK_DESCALE(dx[0] * k.w00 + dx[1] * k.w01 + dx[2] * k.w10 + dx[3] * k.w11, 14);
            const int iy = SGX_LK_DESCALE(dy[0] * k.w00 + dy[1] * k.w01 + dy[2] * k.w10 + dy[3] * k.w11, 14);
            Iw[y * W + x] = (short)iv; dIw[(y * W + x) * 2] = (short)ix; dIw[(y * W + x) * 2 + 1] = (short)iy;
            s11 += (long long)ix * ix; s12 += (long long)ix * iy; s22 += (long long)iy * iy;
        }
        const float A11 = sgx_i64_to_f32(s11) * FLT_SCALE, A12 = sgx_i64_to_f32(s12) * FLT_SCALE, A22 = sgx_i64_to_f32(s22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * W * W);
        if (minEig < min_eig || D < FLT_EPSILON) { if (level == 0) status = 0; continue; }
        D = 1.f / D;
        nextx -= half; nexty -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; j++) {
            const int inx = sgx_floor_f(nextx), iny = sgx_floor_f(nexty);
            if (inx < -W || inx >= w || iny < -W || iny >= h) { if (level == 0) status = 0; break; }
            k = sgx_lk_weights(nextx - inx, nexty - iny);
            long long sb1 = 0, sb2 = 0;
            for (int y = 0; y < W; y++) for (int x = 0; x < W; x++) {
                int pv[4];
                for (int c = 0; c < 4; c++) pv[c] = sgx_lk_px(J, w, h, pitch, inx + x + (c & 1), iny + y + (c >> 1));
                const int diff = SGX_LK_DESCALE(pv[0] * k.w00 + pv[1] * k.w01 + pv[2] * k.w10 + pv[3] * k.w11, 9) - Iw[y * W + x];
                sb1 += (long long)diff * dIw[(y * W + x) * 2]; sb2 += (long long)diff * dIw[(y * W + x) * 2 + 1];
            }
            const SgxLkStep st = sgx_lk_step(A11, A12, A22, D, sb1, sb2);
            nextx += st.dx; nexty += st.dy;
            outx = nextx + half; outy = nexty + half;
            if ((double)st.dx * st.dx + (double)st.dy * st.dy <= eps2) break;
            if (j > 0 && fabs((double)(st.dx + pdx)) < 0.01 && fabs((double)(st.dy + pdy)) < 0.01) { outx -= st.dx * 0.5f; outy -= st.dy * 0.5f; break; }
            pdx = st.dx; pdy = st.dy;
        }
        if (status && level == 0) {
            const int ix = sgx_floor_f(outx - half), iy = sgx_floor_f(outy - half);
            if (ix < -W || ix >= w || iy < -W || iy >= h) status = 0;
        }
    }
    *ox = outx; *oy = outy; *ost = status;
}
SGX_KERNEL(256) k_lk_track(SgxLkGeom g, SgxLkArgs A)
{
    SGX_THREADS_BEGIN(tid)
    int f, kb;
    sgx_lk_decode_block((int)blockIdx.x, A.batch, A.kblocks, f, kb);
    const int kp = kb * 4 + (tid >> 6);
    if ((tid & 63) == 0 && kp < A.n[f] && kp < A.cap) {
        const float *kpt = (const float *)(A.keys + ((size_t)f * A.cap + kp) * 28);
        float ox, oy; uint8_t st;
        sgx_lk_track_point(g, A.cur_img + (size_t)f * g.img_stride, A.prev_img + (size_t)f * g.img_stride, kpt[0], kpt[1], A.max_count, A.eps2, A.min_eig, &ox, &oy, &st);
        A.prev_xy[2 * ((size_t)f * A.cap + kp)] = ox; A.prev_xy[2 * ((size_t)f * A.cap + kp) + 1] = oy;
        if (A.status) A.status[(size_t)f * A.cap + kp] = st;
    }
    SGX_THREADS_END
}
#else
/* sum over the 64 lanes of a wave (no overflow by construction at the call sites); the result is wave-uniform.  Four DPP adds give every lane its
 * 16-lane row total (xor 1, xor 2 inside quads, rotate by 4 and by 8 inside the row); two row broadcasts carry the totals into lane 63. */
SGX_DEV int sgx_wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true);     /* row_ror:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);     /* row_ror:8 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    /* row_bcast:15 into rows 1 and 3 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    /* row_bcast:31 into rows 2 and 3 */
    return __builtin_amdgcn_readlane(v, 63);
}
/* exact wave sum of per-lane int32 partials, returned as the correctly rounded float of the exact integer: the low 16 bits and the signed high
 * part are reduced separately (each fits 32 bits); hi * 65536 + lo is rounded ONCE (fused multiply-add on exact float operands; round 6: was a detour through fp64,
 * four double-rate instructions per sum) — identical to (float)(int64) of the oracle */
SGX_DEV float sgx_wave_sum_f32(int v)
{
    const int lo = sgx_wave_sum_i32(v & 0xFFFF), hi = sgx_wave_sum_i32(v >> 16);
    return fmaf((float)hi, 65536.0f, (float)lo);       /* |hi| <= 2^21, lo < 2^22: both conversions and the product are exact, the fused add rounds the exact integer once */
}
SGX_DEV int sgx_mul24(int a, int b) { return __mul24(a, b); }          /* both operands fit 24 bits at every call site */
/* REFLECT_101 for indices at most one image length outside, clamped (the clamp only ever acts on tile bytes no window uses) */
SGX_DEV int sgx_reflect1(int p, int len) { p = p < 0 ? -p : p; p = p >= len ? 2 * len - 2 - p : p; return min(max(p, 0), len - 1); }

typedef short sgx_i16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short sgx_u16x2 __attribute__((ext_vector_type(2)));
SGX_DEV sgx_i16x2 sgx_as_i16x2(uint32_t v) { return __builtin_bit_cast(sgx_i16x2, v); }
SGX_DEV uint32_t sgx_as_u32(sgx_i16x2 v) { return __builtin_bit_cast(uint32_t, v); }
/* (byte a of {hi:lo}, 0, byte b of {hi:lo}, 0): two pixels widened to a 16-bit pair in one v_perm_b32 */
#define SGX_LK_PAIR(hi, lo, a, b) sgx_as_i16x2(__builtin_amdgcn_perm((hi), (lo), (uint32_t)(a) | 0x0c00u | ((uint32_t)(b) << 16) | 0x0c000000u))
/* (high half of p, low half of q): the pair one 16-bit element further along a row of pairs */
SGX_DEV sgx_i16x2 sgx_lk_next(sgx_i16x2 p, sgx_i16x2 q) { return sgx_as_i16x2(__builtin_amdgcn_alignbit(sgx_as_u32(q), sgx_as_u32(p), 16)); }
#if defined(SGX_LK_NODOT) && !defined(SGX_EMU)      /* diagnostic build (round 6): the two products of a dot as two v_mad_i32_i16 — no instruction of the DOT / matrix pipe in the kernel */
SGX_DEV int sgx_lk_dot2_mad(sgx_i16x2 a, sgx_i16x2 b, int c)
{
    const uint32_t ua = __builtin_bit_cast(uint32_t, a), ub = __builtin_bit_cast(uint32_t, b);
    asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,0,0,0]" : "+v"(c) : "v"(ua), "v"(ub));
    asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,1,0,0]" : "+v"(c) : "v"(ua), "v"(ub));
    return c;
}
#define SGX_LK_DOT2(a, b, c) sgx_lk_dot2_mad((a), (b), (c))
#else
#define SGX_LK_DOT2(a, b, c) __builtin_amdgcn_sdot2((a), (b), (c), false)
#endif
/* the first dot product of a window sample, whose accumulator operand (the sample's start value) must survive.  Rounds 4-6 issued it as inline assembly in the three-address
 * VOP3P form (v_dot2_i32_i16) to spare the v_mov the compiler puts in front of its two-address v_dot2c.  That hid a HAZARD from the compiler: on gfx940 / gfx950 a DOT result
 * needs 3 wait states before a VALU instruction of another opcode reads it (4 before one overwrites it), the compiler's hazard recogniser inserts them only for instructions
 * it can see, and the v_dot2c that followed read its accumulator one wait state after the asm.  No wrong result was ever traced to it (the rare LK difference of round 6 had
 * another cause: compiler-generated packed fp32 beside another wave's bf16 matrix products, profiles/r6_lk_priority_diagnosis.md), but the rule is the hardware's: the builtin costs the v_mov (0.7 % of the kernel) and is handled by the compiler. */
SGX_DEV int sgx_lk_dot2_keep(sgx_i16x2 a, sgx_i16x2 b, int c) { return SGX_LK_DOT2(a, b, c); }

/* stage the ROWS x 36-byte patch of `img` whose top-left corner is (ox, oy) (ox a multiple of 4) into the wave's LDS tile: REFLECT_101 outside the
 * image, aligned dwords wherever the image allows.  Lane -> (row within a group of seven, dword column): 63 lanes move seven tile rows per pass. */
template <int PASSES>
SGX_DEV void sgx_lk_stage(uint32_t *tile, const uint8_t *img, int w, int h, int pitch, int ox, int oy, int rl, int cl)
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int gx = ox + 4 * cl;
    const bool inside = gx >= 0 && gx + 4 <= w;
    if (rl < 7) {
        uint32_t v[PASSES];
        if (inside) {                                               /* all loads of the patch are in flight together */
#pragma unroll
            for (int p = 0; p < PASSES; p++) v[p] = *(const uint32_t *)(img + (size_t)sgx_reflect1(oy + p * 7 + rl, h) * pitch + gx);
        } else {
            const int x0 = sgx_reflect1(gx, w), x1 = sgx_reflect1(gx + 1, w), x2 = sgx_reflect1(gx + 2, w), x3 = sgx_reflect1(gx + 3, w);
#pragma unroll
            for (int p = 0; p < PASSES; p++) {
                const uint8_t *src = img + (size_t)sgx_reflect1(oy + p * 7 + rl, h) * pitch;
                v[p] = (uint32_t)src[x0] | ((uint32_t)src[x1] << 8) | ((uint32_t)src[x2] << 16) | ((uint32_t)src[x3] << 24);
            }
        }
#pragma unroll
        for (int p = 0; p < PASSES; p++) tile[(p * 7 + rl) * (SGX_LK_TILE_PITCH / 4) + cl] = v[p];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
}

#ifdef SGX_DEBUG_TAPS      /* one keypoint per wave: superseded by k_lk_trackN<2>, kept in the tap build as the A/B arm SGX_LK_KPW=1 */
SGX_KERNEL(256) k_lk_track(SgxLkGeom g, SgxLkArgs A)
{
    __shared__ uint32_t tiles[4][35 * SGX_LK_TILE_PITCH / 4 + 5];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int f, kb;
    sgx_lk_decode_block((int)blockIdx.x, A.batch, A.kblocks, f, kb);
    const int kp = kb * 4 + wv;
    if (kp >= A.n[f] || kp >= A.cap) return;                       /* wave-uniform: no workgroup barrier is used below */
    uint32_t *tile = tiles[wv];
    const uint8_t *tile8 = (const uint8_t *)tile;
    const int W = SGX_LK_WIN; const float half = 10.0f, FLT_SCALE = 1.f / (1 << 20);
    const float *kpt = (const float *)(A.keys + ((size_t)f * A.cap + kp) * 28);
    const float kx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kpt[0])));
    const float ky = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kpt[1])));
    const uint8_t *I0 = A.cur_img + (size_t)f * g.img_stride, *J0 = A.prev_img + (size_t)f * g.img_stride;
    const int r = lane / 3, s7 = (lane - 3 * r) * 7;               /* window row, first of the lane's seven columns */
    const int rl = lane / 9, cl = lane - 9 * rl;                   /* staging role: tile row within a pass, dword column */
    const bool active = lane < 63;
    const int lane_off = active ? r * SGX_LK_TILE_PITCH + s7 : 0;
    float outx = 0.f, outy = 0.f; int status = 1;

    for (int level = g.nl - 1; level >= 0; level--) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        const uint8_t *I = I0 + g.ioff[level], *J = J0 + g.ioff[level];
        const float sc = 1.0f / (float)(1 << level);
        float prevx = kx * sc, prevy = ky * sc, nextx, nexty;
        if (level == g.nl - 1) { nextx = prevx; nexty = prevy; } else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;
        prevx -= half; prevy -= half;
        const int ipx = sgx_floor_f(prevx), ipy = sgx_floor_f(prevy);
        if (ipx < -W || ipx >= w || ipy < -W || ipy >= h) { if (level == 0) status = 0; continue; }
        SgxLkWeights k = sgx_lk_weights(prevx - ipx, prevy - ipy);
        sgx_i16x2 W0 = sgx_as_i16x2((uint32_t)k.w00 | ((uint32_t)k.w01 << 16)), W1 = sgx_as_i16x2((uint32_t)k.w10 | ((uint32_t)k.w11 << 16));

        // ---- this lane's seven window samples of I and of its Scharr derivatives (kept in registers for every iteration of the level).  The 24 x 24
        //      patch (window + bilinear neighbour + one-pixel derivative apron) is staged once; calcSharrDeriv is evaluated from it on the fly in packed
        //      16-bit arithmetic: tile rows gy0-1 .. gy0+2, columns gx0-1 .. gx0+8 give the derivatives at the 2 x 8 positions the lane interpolates between.
        int ivb[7]; int ix[7], iy[7];                              /* ivb = 256 - (I sample << 9): the accumulator the J interpolation starts from */
        int s11 = 0, s12 = 0, s22 = 0;
        {
            const int gy0 = ipy + r, gx0 = ipx + s7;
            sgx_i16x2 P[4][5];                                     /* rows gy0-1 .. gy0+2, column pairs (0,1) (2,3) .. (8,9) of columns gx0-1 .. gx0+8 */
            const bool direct = ipx >= 1 && ipx + 25 < w && ipy >= 1 && ipy + 22 < h;      /* patch + apron inside the image (wave-uniform): no reflection anywhere */
            if (direct) {                                          /* the lane's 4 x 12 bytes straight from the image (unaligned loads through L1) */
                const uint8_t *gb = I + (size_t)(active ? gy0 - 1 : ipy) * pitch + (active ? gx0 - 1 : ipx);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t q[3];
                    __builtin_memcpy(q, gb + (size_t)j * pitch, 12);
                    P[j][0] = SGX_LK_PAIR(q[0], q[0], 0, 1); P[j][1] = SGX_LK_PAIR(q[0], q[0], 2, 3);
                    P[j][2] = SGX_LK_PAIR(q[1], q[1], 0, 1); P[j][3] = SGX_LK_PAIR(q[1], q[1], 2, 3);
                    P[j][4] = SGX_LK_PAIR(q[2], q[2], 0, 1);
                }
            } else {                                               /* near the border: REFLECT_101 patch through the LDS tile */
                const int tx = ((ipx - 1) >> 2) << 2, ty = ipy - 1;
                sgx_lk_stage<4>(tile, I, w, h, pitch, tx, ty, rl, cl);
                const uint8_t *tb = tile8 + lane_off + (ipx - 1 - tx);                /* tile row gy0-1, column gx0-1 */
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t q[3];
                    __builtin_memcpy(q, tb + j * SGX_LK_TILE_PITCH, 12);
                    P[j][0] = SGX_LK_PAIR(q[0], q[0], 0, 1); P[j][1] = SGX_LK_PAIR(q[0], q[0], 2, 3);
                    P[j][2] = SGX_LK_PAIR(q[1], q[1], 0, 1); P[j][3] = SGX_LK_PAIR(q[1], q[1], 2, 3);
                    P[j][4] = SGX_LK_PAIR(q[2], q[2], 0, 1);
                }
            }
            const sgx_i16x2 c3 = { 3, 3 }, c10 = { 10, 10 };
            const sgx_i16x2 Wd0 = active ? W0 : sgx_as_i16x2(0u), Wd1 = active ? W1 : sgx_as_i16x2(0u);
            sgx_i16x2 S0[5], S1[5], D0[5], D1[5];                  /* column sums (3,10,3) and differences (-1,0,1) for derivative rows gy0 and gy0+1 */
#pragma unroll
            for (int q = 0; q < 5; q++) {
                S0[q] = (P[0][q] + P[2][q]) * c3 + P[1][q] * c10; D0[q] = P[2][q] - P[0][q];
                S1[q] = (P[1][q] + P[3][q]) * c3 + P[2][q] * c10; D1[q] = P[3][q] - P[1][q];
            }
            // derivative pairs at positions (gx0 + 2q, gx0 + 2q + 1), q = 0..3, rows gy0 (a) and gy0+1 (b)
            sgx_i16x2 xa[5], ya[5], xb[5], yb[5];
            const bool whole = ipx >= 0 && ipx + W < w && ipy >= 0 && ipy + W < h;       /* every position the window touches is inside the image (wave-uniform) */
#pragma unroll
            for (int q = 0; q < 4; q++) {
                xa[q] = S0[q + 1] - S0[q]; ya[q] = (D0[q] + D0[q + 1]) * c3 + sgx_lk_next(D0[q], D0[q + 1]) * c10;
                xb[q] = S1[q + 1] - S1[q]; yb[q] = (D1[q] + D1[q + 1]) * c3 + sgx_lk_next(D1[q], D1[q + 1]) * c10;
            }
            if (!whole) {                                           /* the derivative plane is 0 outside the image (copyMakeBorder CONSTANT) */
                const bool rowa = gy0 >= 0 && gy0 < h, rowb = gy0 + 1 >= 0 && gy0 + 1 < h;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int x = gx0 + 2 * q;
                    const uint32_t m = ((x >= 0 && x < w) ? 0xFFFFu : 0u) | ((x + 1 >= 0 && x + 1 < w) ? 0xFFFF0000u : 0u);
                    const uint32_t ma = rowa ? m : 0u, mb = rowb ? m : 0u;
                    xa[q] = sgx_as_i16x2(sgx_as_u32(xa[q]) & ma); ya[q] = sgx_as_i16x2(sgx_as_u32(ya[q]) & ma);
                    xb[q] = sgx_as_i16x2(sgx_as_u32(xb[q]) & mb); yb[q] = sgx_as_i16x2(sgx_as_u32(yb[q]) & mb);
                }
            }
#pragma unroll
            for (int c = 0; c < 7; c++) {
                // pairs (c, c+1): even c are the stored pairs, odd c straddle two of them
                const sgx_i16x2 pxa = (c & 1) ? sgx_lk_next(xa[c >> 1], xa[(c >> 1) + 1]) : xa[c >> 1], pxb = (c & 1) ? sgx_lk_next(xb[c >> 1], xb[(c >> 1) + 1]) : xb[c >> 1];
                const sgx_i16x2 pya = (c & 1) ? sgx_lk_next(ya[c >> 1], ya[(c >> 1) + 1]) : ya[c >> 1], pyb = (c & 1) ? sgx_lk_next(yb[c >> 1], yb[(c >> 1) + 1]) : yb[c >> 1];
                // image pair at columns (c+1, c+2) of the 10-column rows 1 and 2: odd-aligned for even c
                const sgx_i16x2 pia = (c & 1) ? P[1][(c + 1) >> 1] : sgx_lk_next(P[1][c >> 1], P[1][(c >> 1) + 1]), pib = (c & 1) ? P[2][(c + 1) >> 1] : sgx_lk_next(P[2][c >> 1], P[2][(c >> 1) + 1]);
                const int v = SGX_LK_DOT2(pib, W1, SGX_LK_DOT2(pia, W0, 256)) >> 9;               /* the idle lane's intensity sample is never used: its ix = iy = 0 */
                const int vx = SGX_LK_DOT2(pxb, Wd1, SGX_LK_DOT2(pxa, Wd0, 8192)) >> 14;          /* idle lane: zero weights -> (0 + 8192) >> 14 = 0 */
                const int vy = SGX_LK_DOT2(pyb, Wd1, SGX_LK_DOT2(pya, Wd0, 8192)) >> 14;
                ivb[c] = 256 - (v << 9); ix[c] = vx; iy[c] = vy;
                s11 += sgx_mul24(vx, vx); s12 += sgx_mul24(vx, vy); s22 += sgx_mul24(vy, vy);
            }
        }
        const float A11 = sgx_wave_sum_f32(s11) * FLT_SCALE, A12 = sgx_wave_sum_f32(s12) * FLT_SCALE, A22 = sgx_wave_sum_f32(s22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * W * W);
        if (minEig < A.min_eig || D < FLT_EPSILON) { if (level == 0) status = 0; continue; }
        D = 1.f / D;
        nextx -= half; nexty -= half;
        float pdx = 0.f, pdy = 0.f;
        int ox = 0, oy = 0; bool staged = false;
        for (int j = 0; j < A.max_count; j++) {
            const int inx = sgx_floor_f(nextx), iny = sgx_floor_f(nexty);
            if (inx < -W || inx >= w || iny < -W || iny >= h) { if (level == 0) status = 0; break; }
            k = sgx_lk_weights(nextx - inx, nexty - iny);
            W0 = sgx_as_i16x2((uint32_t)k.w00 | ((uint32_t)k.w01 << 16)); W1 = sgx_as_i16x2((uint32_t)k.w10 | ((uint32_t)k.w11 << 16));
            if (!staged || inx < ox || inx - ox > SGX_LK_TILE_PITCH - 22 || iny < oy || iny - oy > 35 - 22) {
                ox = ((inx - 5) >> 2) << 2; oy = iny - 6;           /* five to eight pixels of room on every side before the window leaves the tile */
                sgx_lk_stage<5>(tile, J, w, h, pitch, ox, oy, rl, cl);
                staged = true;
            }
            int sb1 = 0, sb2 = 0;
            {
                const uint8_t *tb = tile8 + lane_off + (iny - oy) * SGX_LK_TILE_PITCH + (inx - ox);
                uint32_t a[2], b[2];
                __builtin_memcpy(a, tb, 8); __builtin_memcpy(b, tb + SGX_LK_TILE_PITCH, 8);
#pragma unroll
                for (int c = 0; c < 7; c++) {
                    const int diff = SGX_LK_DOT2(SGX_LK_PAIR(b[1], b[0], c, c + 1), W1, SGX_LK_DOT2(SGX_LK_PAIR(a[1], a[0], c, c + 1), W0, ivb[c])) >> 9;
                    sb1 += sgx_mul24(diff, ix[c]); sb2 += sgx_mul24(diff, iy[c]);          /* ix = iy = 0 on the idle lane */
                }
            }
            const float b1 = sgx_wave_sum_f32(sb1) * FLT_SCALE, b2 = sgx_wave_sum_f32(sb2) * FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
            nextx += dx; nexty += dy;
            outx = nextx + half; outy = nexty + half;
            if ((double)dx * dx + (double)dy * dy <= A.eps2) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) { outx -= dx * 0.5f; outy -= dy * 0.5f; break; }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {
            const int qx = sgx_floor_f(outx - half), qy = sgx_floor_f(outy - half);
            if (qx < -W || qx >= w || qy < -W || qy >= h) status = 0;
        }
    }
    if (lane == 0) {
        A.prev_xy[2 * ((size_t)f * A.cap + kp)] = outx; A.prev_xy[2 * ((size_t)f * A.cap + kp) + 1] = outy;
        if (A.status) A.status[(size_t)f * A.cap + kp] = (uint8_t)status;
    }
}
#endif      /* SGX_DEBUG_TAPS */
#endif

// ---------------------------------------------------------------------------------------------
// k_lk_trackN<KPW>: the same tracker with KPW = 2 or 4 keypoints per wave (one per group of LPK = 64 / KPW lanes: a half-wave, or a 16-lane DPP row).  The 63 seven-sample
// segments of a window are dealt to the LPK lanes of a group (segments l, l + LPK, ...), so a lane does per keypoint what KPW lanes of k_lk_track do — the per-sample work per keypoint is unchanged — while everything
// that k_lk_track pays once per wave is now paid once per KPW keypoints: the cross-lane reductions (4 DPP adds inside a row, plus one lane-xor-16 exchange for half-waves, instead of 6 + a readlane),
// the float step arithmetic, the window / tile bookkeeping.  The groups of a wave iterate together; a finished group idles (its lanes are masked) until the slowest
// one stops (measured on the bench streams: 13.1 iterations per keypoint on average, 14.6 / 16.1 for the maximum over 2 / 4 neighbours).  KPW = 2 needs half the
// registers of KPW = 4 (two segments per lane instead of four) and runs at twice the occupancy.  All sums are the same exact integers, so the results are bit-identical to k_lk_track and to the oracle's exact-sum mode.
// Workgroup = 4 waves = 4 KPW keypoints; LDS: one 36 x 32-byte tile per keypoint.
// ---------------------------------------------------------------------------------------------
#ifndef SGX_EMU
SGX_DEV int sgx_row_sum_i32(int v)                                  /* every lane receives the total of its 16-lane row */
{
    /* quad permutes and row rotations read a valid lane everywhere, so the `old` operand is never used: passing v itself spares the v_mov 0 the compiler
     * would otherwise emit in front of every DPP add */
    v += __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, true);      /* quad_perm [1,0,3,2] */
    v += __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, true);      /* quad_perm [2,3,0,1] */
    v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, true);     /* row_ror:4 */
    v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, true);     /* row_ror:8 */
    return v;
}
template <int LPK>
SGX_DEV float sgx_group_sum_f32(int v)                              /* exact sum over the LPK lanes of a group -> correctly rounded float, as sgx_wave_sum_f32 */
{
    int lo = sgx_row_sum_i32(v & 0xFFFF), hi = sgx_row_sum_i32(v >> 16);
    if (LPK == 32) { lo += __shfl_xor(lo, 16, 64); hi += __shfl_xor(hi, 16, 64); }      /* the other row of the half-wave */
    return fmaf((float)hi, 65536.0f, (float)lo);       /* |hi| <= 2^21, lo < 2^22: both conversions and the product are exact, the fused add rounds the exact integer once */
}

/* stage the NROWS x 32-byte patch of `img` whose top-left corner is (ox, oy) (ox a multiple of 4) into a keypoint's tile with the LPK lanes of its group (LPK / 8 tile rows of
 * eight dwords per pass): REFLECT_101 outside the image, aligned dwords wherever the image allows. */
#define SGX_LK4_PITCH 32
template <int NROWS, int LPK>
SGX_DEV void sgx_lk_stage_row(uint32_t *tile, const uint8_t *img, int w, int h, int pitch, int ox, int oy, int l16)
{
    constexpr int RPP = LPK / 8, G = LPK == 16 ? 6 : 3;              /* tile rows per pass; loads in flight per lane before their LDS stores */
    const int c = l16 & 7, rs = l16 >> 3, gx = ox + 4 * c;
    const bool inside = gx >= 0 && gx + 4 <= w;
    const int x0 = sgx_reflect1(gx, w), x1 = sgx_reflect1(gx + 1, w), x2 = sgx_reflect1(gx + 2, w), x3 = sgx_reflect1(gx + 3, w);
#pragma unroll 1
    for (int r0 = rs; r0 < NROWS; r0 += RPP * G) {
        uint32_t v[G];
#pragma unroll
        for (int p = 0; p < G; p++) {
            const uint8_t *src = img + (size_t)sgx_reflect1(oy + min(r0 + RPP * p, NROWS - 1), h) * pitch;
            if (inside) v[p] = *(const uint32_t *)(src + gx);
            else v[p] = (uint32_t)src[x0] | ((uint32_t)src[x1] << 8) | ((uint32_t)src[x2] << 16) | ((uint32_t)src[x3] << 24);
        }
#pragma unroll
        for (int p = 0; p < G; p++) { const int r = r0 + RPP * p; if (r < NROWS) tile[r * (SGX_LK4_PITCH / 4) + c] = v[p]; }
    }
}

/* one seven-sample segment of the tracked-from window: the packed pixel pairs P (rows gy0-1 .. gy0+2, columns gx0-1 .. gx0+8) -> window intensities (as the
 * accumulator start 256 - (I << 9)), Scharr derivatives, and the segment's share of the gradient matrix.  Same arithmetic as k_lk_track. */
SGX_DEV void sgx_lk_segment_setup(const sgx_i16x2 (&P)[4][5], sgx_i16x2 W0, sgx_i16x2 W1, bool whole, int gx0, int gy0, int w, int h,
                                  int (&ivb)[7], uint32_t (&ixy)[7], int &s11, int &s12, int &s22)
{
    const sgx_i16x2 c3 = { 3, 3 }, c10 = { 10, 10 };
    sgx_i16x2 S0[5], S1[5], D0[5], D1[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
        S0[q] = (P[0][q] + P[2][q]) * c3 + P[1][q] * c10; D0[q] = P[2][q] - P[0][q];
        S1[q] = (P[1][q] + P[3][q]) * c3 + P[2][q] * c10; D1[q] = P[3][q] - P[1][q];
    }
    sgx_i16x2 xa[5], ya[5], xb[5], yb[5];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        xa[q] = S0[q + 1] - S0[q]; ya[q] = (D0[q] + D0[q + 1]) * c3 + sgx_lk_next(D0[q], D0[q + 1]) * c10;
        xb[q] = S1[q + 1] - S1[q]; yb[q] = (D1[q] + D1[q + 1]) * c3 + sgx_lk_next(D1[q], D1[q + 1]) * c10;
    }
    if (!whole) {                                                   /* the derivative plane is 0 outside the image (copyMakeBorder CONSTANT) */
        const bool rowa = gy0 >= 0 && gy0 < h, rowb = gy0 + 1 >= 0 && gy0 + 1 < h;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int x = gx0 + 2 * q;
            const uint32_t m = ((x >= 0 && x < w) ? 0xFFFFu : 0u) | ((x + 1 >= 0 && x + 1 < w) ? 0xFFFF0000u : 0u);
            const uint32_t ma = rowa ? m : 0u, mb = rowb ? m : 0u;
            xa[q] = sgx_as_i16x2(sgx_as_u32(xa[q]) & ma); ya[q] = sgx_as_i16x2(sgx_as_u32(ya[q]) & ma);
            xb[q] = sgx_as_i16x2(sgx_as_u32(xb[q]) & mb); yb[q] = sgx_as_i16x2(sgx_as_u32(yb[q]) & mb);
        }
    }
#pragma unroll
    for (int c = 0; c < 7; c++) {
        const sgx_i16x2 pxa = (c & 1) ? sgx_lk_next(xa[c >> 1], xa[(c >> 1) + 1]) : xa[c >> 1], pxb = (c & 1) ? sgx_lk_next(xb[c >> 1], xb[(c >> 1) + 1]) : xb[c >> 1];
        const sgx_i16x2 pya = (c & 1) ? sgx_lk_next(ya[c >> 1], ya[(c >> 1) + 1]) : ya[c >> 1], pyb = (c & 1) ? sgx_lk_next(yb[c >> 1], yb[(c >> 1) + 1]) : yb[c >> 1];
        const sgx_i16x2 pia = (c & 1) ? P[1][(c + 1) >> 1] : sgx_lk_next(P[1][c >> 1], P[1][(c >> 1) + 1]), pib = (c & 1) ? P[2][(c + 1) >> 1] : sgx_lk_next(P[2][c >> 1], P[2][(c >> 1) + 1]);
        const int v = SGX_LK_DOT2(pib, W1, SGX_LK_DOT2(pia, W0, 256)) >> 9;
        const int vx = SGX_LK_DOT2(pxb, W1, SGX_LK_DOT2(pxa, W0, 8192)) >> 14;
        const int vy = SGX_LK_DOT2(pyb, W1, SGX_LK_DOT2(pya, W0, 8192)) >> 14;
        ivb[c] = 256 - (v << 9); ixy[c] = ((uint32_t)vx & 0xFFFFu) | ((uint32_t)vy << 16);          /* |vx|, |vy| <= 16 * 255: they fit 16 bits */
        s11 += sgx_mul24(vx, vx); s12 += sgx_mul24(vx, vy); s22 += sgx_mul24(vy, vy);
    }
}
/* acc += diff * (low / high signed half of pair): v_mad_i32_i16 with op_sel picks the half, so the (Ix, Iy) pairs stay packed in one register per sample */
SGX_DEV int sgx_mad_lo16(int diff, uint32_t pair, int acc) { asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,0,0,0]" : "+v"(acc) : "v"(diff), "v"(pair)); return acc; }
SGX_DEV int sgx_mad_hi16(int diff, uint32_t pair, int acc) { asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(acc) : "v"(diff), "v"(pair)); return acc; }

template <int KPW>
SGX_KERNEL_OCC(256, (KPW == 4 ? 3 : 5)) k_lk_trackN(SgxLkGeom g, SgxLkArgs A)
{
    constexpr int LPK = 64 / KPW, SPL = KPW;                         /* lanes per keypoint, segments per lane (SPL * LPK = 64 slots for 63 segments) */
    constexpr int TD = 36 * SGX_LK4_PITCH / 4 + 5;                 /* dwords per keypoint tile: 36 rows of 32 bytes (+ slack for the 8-byte reads past the last row; odd stride between tiles) */
    __shared__ uint32_t tiles[4 * KPW][TD];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane / LPK, l16 = lane % LPK;
    int f, kb;
    sgx_lk_decode_block((int)blockIdx.x, A.batch, A.kblocks, f, kb);
    const int nkp = min(A.n[f], A.cap);
    if (kb * (4 * KPW) + wv * KPW >= nkp) return;                            /* wave-uniform: no workgroup barrier is used below */
    const int kp = kb * (4 * KPW) + wv * KPW + row;
    const bool valid = kp < nkp;
    uint32_t *tile = tiles[wv * KPW + row];
    const uint8_t *tile8 = (const uint8_t *)tile;
    const int W = SGX_LK_WIN; const float half = 10.0f, FLT_SCALE = 1.f / (1 << 20);
    const float *kpt = (const float *)(A.keys + ((size_t)f * A.cap + (valid ? kp : 0)) * 28);
    const float kx = kpt[0], ky = kpt[1];
    const uint8_t *I0 = A.cur_img + (size_t)f * g.img_stride, *J0 = A.prev_img + (size_t)f * g.img_stride;
    // the lane's SPL segments: segment index l16 + LPK q -> (window row, first of seven columns); the 64th slot (last lane, last q) is idle: zero weights
    int soff[SPL], srow[SPL], scol[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) { const int si = min(l16 + LPK * q, 62); srow[q] = si / 3; scol[q] = (si - 3 * srow[q]) * 7; soff[q] = srow[q] * SGX_LK4_PITCH + scol[q]; }
    const bool idle3 = l16 == LPK - 1;                              /* the last slot of the last lane does not exist */
    float outx = 0.f, outy = 0.f; int status = 1;

    for (int level = g.nl - 1; level >= 0; level--) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        const uint8_t *I = I0 + g.ioff[level], *J = J0 + g.ioff[level];
        const float sc = 1.0f / (float)(1 << level);
        float prevx = kx * sc, prevy = ky * sc, nextx, nexty;
        if (level == g.nl - 1) { nextx = prevx; nexty = prevy; } else { nextx = outx * 2.f; nexty = outy * 2.f; }
        outx = nextx; outy = nexty;
        prevx -= half; prevy -= half;
        const int ipx = sgx_floor_f(prevx), ipy = sgx_floor_f(prevy);
        bool act = valid;                                           /* this row still works on this level */
        if (ipx < -W || ipx >= w || ipy < -W || ipy >= h) { if (level == 0) status = 0; act = false; }
        if (!__any(act)) continue;
        SgxLkWeights k = sgx_lk_weights(prevx - ipx, prevy - ipy);
        sgx_i16x2 W0 = sgx_as_i16x2((uint32_t)k.w00 | ((uint32_t)k.w01 << 16)), W1 = sgx_as_i16x2((uint32_t)k.w10 | ((uint32_t)k.w11 << 16));

        int ivb[SPL][7]; uint32_t ixy[SPL][7];
        int s11 = 0, s12 = 0, s22 = 0;
        {
            const bool direct = act && ipx >= 1 && ipx + 25 < w && ipy >= 1 && ipy + 22 < h;       /* patch + apron inside the image: no reflection anywhere (per row) */
            const bool whole = !__any(act && !(ipx >= 0 && ipx + W < w && ipy >= 0 && ipy + W < h));      /* wave-uniform: no active window of this wave touches the border -> the masking code is skipped, not predicated */
            const int tx = ((ipx - 1) >> 2) << 2, ty = ipy - 1;
            if (__any(act && !direct)) {                            /* near the border: REFLECT_101 patch through the keypoint's LDS tile */
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (act && !direct) sgx_lk_stage_row<28, LPK>(tile, I, w, h, pitch, tx, ty, l16);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int q = 0; q < SPL; q++) {
                const int gy0 = ipy + srow[q], gx0 = ipx + scol[q];
                sgx_i16x2 P[4][5];
                if (direct) {
                    const uint8_t *gb = I + (size_t)(gy0 - 1) * pitch + (gx0 - 1);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t d[3];
                        __builtin_memcpy(d, gb + (size_t)j * pitch, 12);
                        P[j][0] = SGX_LK_PAIR(d[0], d[0], 0, 1); P[j][1] = SGX_LK_PAIR(d[0], d[0], 2, 3);
                        P[j][2] = SGX_LK_PAIR(d[1], d[1], 0, 1); P[j][3] = SGX_LK_PAIR(d[1], d[1], 2, 3);
                        P[j][4] = SGX_LK_PAIR(d[2], d[2], 0, 1);
                    }
                } else {
                    const uint8_t *tb = tile8 + soff[q] + (ipx - 1 - tx);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t d[3];
                        __builtin_memcpy(d, tb + j * SGX_LK4_PITCH, 12);
                        P[j][0] = SGX_LK_PAIR(d[0], d[0], 0, 1); P[j][1] = SGX_LK_PAIR(d[0], d[0], 2, 3);
                        P[j][2] = SGX_LK_PAIR(d[1], d[1], 0, 1); P[j][3] = SGX_LK_PAIR(d[1], d[1], 2, 3);
                        P[j][4] = SGX_LK_PAIR(d[2], d[2], 0, 1);
                    }
                }
                const bool on = act && !(q == SPL - 1 && idle3);          /* inactive rows and the idle slot: zero weights -> zero samples */
                const sgx_i16x2 Wq0 = on ? W0 : sgx_as_i16x2(0u), Wq1 = on ? W1 : sgx_as_i16x2(0u);
                sgx_lk_segment_setup(P, Wq0, Wq1, whole, gx0, gy0, w, h, ivb[q], ixy[q], s11, s12, s22);
                __builtin_amdgcn_sched_barrier(0);                  /* one segment at a time: keeps the 60 temporaries of a segment from overlapping the next one's */
            }
        }
        const float A11 = sgx_group_sum_f32<LPK>(s11) * FLT_SCALE, A12 = sgx_group_sum_f32<LPK>(s12) * FLT_SCALE, A22 = sgx_group_sum_f32<LPK>(s22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * W * W);
        if (act && (minEig < A.min_eig || D < FLT_EPSILON)) { if (level == 0) status = 0; act = false; }
        D = 1.f / D;
        nextx -= half; nexty -= half;
        float pdx = 0.f, pdy = 0.f;
        int ox = 0, oy = 0; bool staged = false;
        for (int j = 0; j < A.max_count; j++) {
            const int inx = sgx_floor_f(nextx), iny = sgx_floor_f(nexty);
            if (act && (inx < -W || inx >= w || iny < -W || iny >= h)) { if (level == 0) status = 0; act = false; }
            if (!__any(act)) break;
            k = sgx_lk_weights(nextx - inx, nexty - iny);
            W0 = sgx_as_i16x2((uint32_t)k.w00 | ((uint32_t)k.w01 << 16)); W1 = sgx_as_i16x2((uint32_t)k.w10 | ((uint32_t)k.w11 << 16));
            const bool need = act && (!staged || inx < ox || inx - ox > SGX_LK4_PITCH - 22 || iny < oy || iny - oy > 36 - 22);
            if (__any(need)) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (need) {
                    ox = ((inx - 3) >> 2) << 2; oy = iny - 7;       /* three to six pixels of room left and right, seven above and below, before the window leaves the tile */
                    sgx_lk_stage_row<36, LPK>(tile, J, w, h, pitch, ox, oy, l16);
                    staged = true;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
            int sb1 = 0, sb2 = 0;
            if (act) {
                const uint8_t *tb0 = tile8 + (iny - oy) * SGX_LK4_PITCH + (inx - ox);
#pragma unroll
                for (int q = 0; q < SPL; q++) {
                    const uint8_t *tb = tb0 + soff[q];
                    uint32_t a[2], b[2];
                    __builtin_memcpy(a, tb, 8); __builtin_memcpy(b, tb + SGX_LK4_PITCH, 8);
#pragma unroll
                    for (int c = 0; c < 7; c++) {
                        const int diff = SGX_LK_DOT2(SGX_LK_PAIR(b[1], b[0], c, c + 1), W1, sgx_lk_dot2_keep(SGX_LK_PAIR(a[1], a[0], c, c + 1), W0, ivb[q][c])) >> 9;
                        sb1 = sgx_mad_lo16(diff, ixy[q][c], sb1); sb2 = sgx_mad_hi16(diff, ixy[q][c], sb2);      /* |diff| <= 255 * 32 fits 16 bits; ix = iy = 0 on the idle slot */
                    }
                }
            }
            const float b1 = sgx_group_sum_f32<LPK>(sb1) * FLT_SCALE, b2 = sgx_group_sum_f32<LPK>(sb2) * FLT_SCALE;
            if (act) {
                const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
                nextx += dx; nexty += dy;
                outx = nextx + half; outy = nexty + half;
                // delta.ddot(delta) <= epsilon is a DOUBLE comparison in lkpyramid.cpp.  The float estimate f2 is within 2e-7 (relative) of the double value, so outside a 1e-5 band
                // around the threshold it decides the same way; inside the band (any lane of the wave: wave-uniform branch) the double expression itself decides.  Round 6: the
                // five double-rate instructions ran in every iteration of every lane.
                const float f2 = dx * dx + dy * dy, e2f = (float)A.eps2;
                bool conv = f2 <= e2f;
                if (__any(fabsf(f2 - e2f) <= 1e-5f * e2f)) { asm volatile("" ::: "memory"); conv = (double)dx * dx + (double)dy * dy <= A.eps2; }      /* (the empty asm keeps the compiler from speculating the double path) */
                if (conv) act = false;
                // fabs((double)(float sum)) < 0.01  <=>  |sum| <= 0.01f: 0.01f = 0.00999999977648 is the largest float below the double 0.01
                else if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) { outx -= dx * 0.5f; outy -= dy * 0.5f; act = false; }
                pdx = dx; pdy = dy;
            }
        }
        if (valid && status && level == 0) {
            const int qx = sgx_floor_f(outx - half), qy = sgx_floor_f(outy - half);
            if (qx < -W || qx >= w || qy < -W || qy >= h) status = 0;
        }
    }
    if (l16 == 0 && valid) {
        A.prev_xy[2 * ((size_t)f * A.cap + kp)] = outx; A.prev_xy[2 * ((size_t)f * A.cap + kp) + 1] = outy;
        if (A.status) A.status[(size_t)f * A.cap + kp] = (uint8_t)status;
    }
}
#else
template <int KPW>
SGX_KERNEL(256) k_lk_trackN(SgxLkGeom g, SgxLkArgs A)
{
    SGX_THREADS_BEGIN(tid)
    int f, kb;
    sgx_lk_decode_block((int)blockIdx.x, A.batch, A.kblocks, f, kb);
    const int kp = kb * (4 * KPW) + tid / (64 / KPW);
    if (tid % (64 / KPW) == 0 && kp < A.n[f] && kp < A.cap) {
        const float *kpt = (const float *)(A.keys + ((size_t)f * A.cap + kp) * 28);
        float ox, oy; uint8_t st;
        sgx_lk_track_point(g, A.cur_img + (size_t)f * g.img_stride, A.prev_img + (size_t)f * g.img_stride, kpt[0], kpt[1], A.max_count, A.eps2, A.min_eig, &ox, &oy, &st);
        A.prev_xy[2 * ((size_t)f * A.cap + kp)] = ox; A.prev_xy[2 * ((size_t)f * A.cap + kp) + 1] = oy;
        if (A.status) A.status[(size_t)f * A.cap + kp] = st;
    }
    SGX_THREADS_END
}
#endif

// ---------------------------------------------------------------------------------------------
// findFundamentalMat(FM_RANSAC)
// ---------------------------------------------------------------------------------------------
#define SGX_FM_CHUNK 32               /* 7-index groups drawn per round */
#define SGX_FM_MAXPTS 2048

/* cv::solveCubic (mathfuncs.cpp) */
SGX_DEV int sgx_solve_cubic(const double *c, double *roots)
{
    double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
    double x0 = 0., x1 = 0., x2 = 0.;
    int n = 0;
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) n = a3 == 0 ? -1 : 0;
            else { x0 = -a3 / a2; n = 1; }
        } else {
            double d = a2 * a2 - 4 * a1 * a3;
            if (d >= 0) {
                d = sqrt(d);
                double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
                if (fabs(q1) > fabs(q2)) { x0 = q1 / a1; x1 = a3 / q1; }
                else { x0 = q2 / a1; x1 = a3 / q2; }
                n = d > 0 ? 2 : 1;
            }
        }
    } else {
        a0 = 1. / a0; a1 *= a0; a2 *= a0; a3 *= a0;
        double Q = (a1 * a1 - 3 * a2) * (1. / 9);
        double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
        double Qcubed = Q * Q * Q;
        double d = Qcubed - R * R;
        if (d > 0) {
            double theta = acos(R / sqrt(Qcubed));
            double sqrtQ = sqrt(Q);
            double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
            x0 = t0 * cos(t1) - t2;
            x1 = t0 * cos(t1 + (2. * 3.1415926535897932384626433832795 / 3)) - t2;
            x2 = t0 * cos(t1 + (4. * 3.1415926535897932384626433832795 / 3)) - t2;
            n = 3;
        } else if (d == 0) {
            if (R >= 0) { x0 = -2 * pow(R, 1. / 3) - a1 / 3; x1 = pow(R, 1. / 3) - a1 / 3; }
            else { x0 = 2 * pow(-R, 1. / 3) - a1 / 3; x1 = -pow(-R, 1. / 3) - a1 / 3; }
            x2 = 0;
            n = x0 == x1 ? 1 : 2;
            x1 = x0 == x1 ? 0 : x1;
        } else {
            double e;
            d = sqrt(-d);
            e = pow(d + fabs(R), 1. / 3);
            if (R > 0) e = -e;
            x0 = (e + Q / e) - a1 * (1. / 3);
            n = 1;
        }
    }
    roots[0] = x0; roots[1] = x1; roots[2] = x2;
    return n;
}

/* run7Point (fundam.cpp).  M = 63-double work area (A^T, 9 x 7, overwritten by the Householder vectors).  Null space of the 7 x 9 system =
 * the last two columns of Q in A^T = QR: q_j = H_0 H_1 ... H_6 e_j (same arithmetic as oracle/flow_oracle.c null_space_7x9). */
SGX_DEV int sgx_fm_run7point(const float *m1, const float *m2, double *M, double *fmatrix)
{
    double beta[7], f1[9], f2[9], c[4], r[3];
    for (int i = 0; i < 7; i++) {
        const double x0 = m1[2 * i], y0 = m1[2 * i + 1], x1 = m2[2 * i], y1 = m2[2 * i + 1];
        M[0 * 7 + i] = x1 * x0; M[1 * 7 + i] = x1 * y0; M[2 * 7 + i] = x1;
        M[3 * 7 + i] = y1 * x0; M[4 * 7 + i] = y1 * y0; M[5 * 7 + i] = y1;
        M[6 * 7 + i] = x0; M[7 * 7 + i] = y0; M[8 * 7 + i] = 1;
    }
    for (int k = 0; k < 7; k++) {
        double nrm = 0; for (int i = k; i < 9; i++) nrm += M[i * 7 + k] * M[i * 7 + k];
        nrm = sqrt(nrm);
        beta[k] = 0;
        if (nrm == 0) continue;
        M[k * 7 + k] += M[k * 7 + k] >= 0 ? nrm : -nrm;                 /* column k, rows k..8 now hold v_k */
        double vv = 0; for (int i = k; i < 9; i++) vv += M[i * 7 + k] * M[i * 7 + k];
        if (vv == 0) continue;
        beta[k] = 2. / vv;
        for (int j = k + 1; j < 7; j++) {
            double s = 0; for (int i = k; i < 9; i++) s += M[i * 7 + k] * M[i * 7 + j];
            s *= beta[k];
            for (int i = k; i < 9; i++) M[i * 7 + j] -= s * M[i * 7 + k];
        }
    }
    for (int i = 0; i < 9; i++) { f1[i] = i == 7; f2[i] = i == 8; }
    for (int k = 6; k >= 0; k--) {
        double s1 = 0, s2 = 0;
        for (int i = k; i < 9; i++) { s1 += M[i * 7 + k] * f1[i]; s2 += M[i * 7 + k] * f2[i]; }
        s1 *= beta[k]; s2 *= beta[k];
        for (int i = k; i < 9; i++) { f1[i] -= s1 * M[i * 7 + k]; f2[i] -= s2 * M[i * 7 + k]; }
    }
    for (int i = 0; i < 9; i++) f1[i] -= f2[i];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 -
           f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) - f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) +
           f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) + f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7]; t1 = f1[3] * f1[8] - f1[5] * f1[6]; t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 -
           f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) - f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) +
           f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) + f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    const int n = sgx_solve_cubic(c, r);
    if (n < 1 || n > 3) return n;
    for (int k = 0; k < n; k++, fmatrix += 9) {
        double lambda = r[k], mu = 1.;
        const double s = f1[8] * r[k] + f2[8];
        if (fabs(s) > DBL_EPSILON) { mu = 1. / s; lambda *= mu; fmatrix[8] = 1.; }
        else fmatrix[8] = 0.;
        for (int i = 0; i < 8; i++) fmatrix[i] = f1[i] * lambda + f2[i] * mu;
    }
    return n;
}

/* FMEstimatorCallback::computeError for one pair, as float */
SGX_DEV float sgx_fm_error(const double *F, float m1x, float m1y, float m2x, float m2y)
{
    double a, b, c, d1, d2, s1, s2;
    a = F[0] * m1x + F[1] * m1y + F[2];
    b = F[3] * m1x + F[4] * m1y + F[5];
    c = F[6] * m1x + F[7] * m1y + F[8];
    s2 = 1. / (a * a + b * b);
    d2 = m2x * a + m2y * b + c;
    a = F[0] * m2x + F[3] * m2y + F[6];
    b = F[1] * m2x + F[4] * m2y + F[7];
    c = F[2] * m2x + F[5] * m2y + F[8];
    s1 = 1. / (a * a + b * b);
    d1 = m1x * a + m1y * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 < e2 ? e2 : e1);                               /* std::max(e1, e2) */
}

/* haveCollinearPoints(m, 7): does the LAST point lie on a line through two earlier ones (ptsetreg.cpp) */
SGX_DEV bool sgx_fm_collinear7(const float *p)
{
    const int i = 6;
    for (int j = 0; j < i; j++) {
        const double dx1 = p[2 * j] - p[2 * i], dy1 = p[2 * j + 1] - p[2 * i + 1];
        for (int k = 0; k < j; k++) {
            const double dx2 = p[2 * k] - p[2 * i], dy2 = p[2 * k + 1] - p[2 * i + 1];
            if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
        }
    }
    return false;
}

/* RANSACUpdateNumIters (ptsetreg.cpp) */
SGX_DEV int sgx_ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

struct SgxFmArgs {
    const uint8_t *keys; const int *n; int cap; const float *prev_xy;
    const int *pre_have; const float *pre_boxes; const int *pre_nboxes; int max_boxes;       /* previous frame's person boxes (may be NULL) */
    double threshold, confidence; int max_iters;
    double *F; int *ok; int *stats;          /* F: 9 doubles per frame; ok: 1 / 0 (no model; F zeroed); stats (optional): 4 ints per frame */
};

#ifndef SGX_FM_OCC
#define SGX_FM_OCC 2      /* waves per SIMD the register allocation is held to: at 243 + 16 registers (round 3) one workgroup = one wave per SIMD had a CU to itself */
#endif
SGX_KERNEL_OCC(256, SGX_FM_OCC) k_fm_ransac(SgxFmArgs A)
{
    SGX_DYN_LDS(lds_raw);
    // layout: m1 (cap float2) | m2 (cap float2) | scan (256 int) | flags (cap u8, padded)
    const int f = (int)blockIdx.x, cap = A.cap;
    float *m1 = (float *)lds_raw, *m2 = m1 + 2 * cap;
    int *scan = (int *)(m2 + 2 * cap);
    uint8_t *flag = (uint8_t *)(scan + 256);
    SGX_LDS int s_total, s_count, s_done, s_niters, s_iter, s_maxgood, s_best_iter, s_best_root, s_attempts;
    SGX_LDS unsigned long long s_rng;
    SGX_LDS int g_idx[SGX_FM_CHUNK][7];
    SGX_LDS int g_nmodels[SGX_FM_CHUNK];            /* -1 = group rejected by checkSubset */
    SGX_LDS double g_model[SGX_FM_CHUNK][27];
    SGX_LDS int g_good[SGX_FM_CHUNK][3];
    SGX_LDS double g_work[SGX_FM_CHUNK][63];
    SGX_LDS double s_best[9];
    SGX_LDS float g_med[SGX_FM_CHUNK][3];           /* LMedS: median error of every model of the round */
    SGX_LDS double s_minmed;
    const int N = min(A.n[f], cap), CH = (N + 255) / 256;
    const bool pre = A.pre_have && A.pre_have[f] && A.pre_boxes && A.pre_nboxes;

    // ---- Frame.cc:454-467: pairs whose previous position is outside every person box of the previous frame
    SGX_THREADS_BEGIN(tid)
    int c = 0;
    for (int i = tid * CH; i < min(N, (tid + 1) * CH); i++) {
        uint8_t keep = 1;
        if (pre) {
            const float x = A.prev_xy[2 * ((size_t)f * cap + i)], y = A.prev_xy[2 * ((size_t)f * cap + i) + 1];
            const int nb = min(A.pre_nboxes[f], A.max_boxes);
            for (int q = 0; q < nb; q++) {
                const float *r = A.pre_boxes + 4 * ((size_t)f * A.max_boxes + q);
                if (x > r[0] && x < r[0] + r[2] && y > r[1] && y < r[1] + r[3]) { keep = 0; break; }
            }
        }
        flag[i] = keep; c += keep;
    }
    scan[tid] = c;
    if (tid == 0) { s_done = 0; s_niters = A.max_iters > 1 ? A.max_iters : 1; s_iter = 0; s_maxgood = 0; s_best_iter = 0; s_best_root = 0; s_attempts = 0; s_rng = ~0ull; }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    sgx_block_exclusive_scan_i32(scan, 256, &s_total, tid);
    SGX_THREADS_END
    SGX_SYNC();
    const bool use_sel = pre && s_total > 20;                       /* :469 */
    SGX_THREADS_BEGIN(tid)
    int pos = use_sel ? scan[tid] : tid * CH;
    for (int i = tid * CH; i < min(N, (tid + 1) * CH); i++) {
        if (use_sel && !flag[i]) continue;
        const float *kp = (const float *)(A.keys + ((size_t)f * cap + i) * 28);
        m1[2 * pos] = kp[0]; m1[2 * pos + 1] = kp[1];
        m2[2 * pos] = A.prev_xy[2 * ((size_t)f * cap + i)]; m2[2 * pos + 1] = A.prev_xy[2 * ((size_t)f * cap + i) + 1];
        pos++;
    }
    if (tid == 0) s_count = use_sel ? s_total : N;
    SGX_THREADS_END
    SGX_SYNC();
    const int count = s_count;
    const float t = (float)(A.threshold * A.threshold);

    if (count < 7) {                                                 /* empty Mat */
        SGX_THREADS_BEGIN(tid)
        if (tid < 9) A.F[9 * (size_t)f + tid] = 0.;
        if (tid == 0) { A.ok[f] = 0; if (A.stats) { A.stats[4 * f] = 0; A.stats[4 * f + 1] = 0; A.stats[4 * f + 2] = 0; A.stats[4 * f + 3] = 0; } }
        SGX_THREADS_END
        return;
    }
    if (count == 7) {                                                /* run7Point on the seven pairs; a 3x3 read of the 9x3 result sees the first root */
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {
            const int k = sgx_fm_run7point(m1, m2, g_work[0], g_model[0]);
            for (int i = 0; i < 9; i++) A.F[9 * (size_t)f + i] = k > 0 ? g_model[0][i] : 0.;
            A.ok[f] = k > 0;
            if (A.stats) { A.stats[4 * f] = 0; A.stats[4 * f + 1] = 0; A.stats[4 * f + 2] = 0; A.stats[4 * f + 3] = 0; }
        }
        SGX_THREADS_END
        return;
    }

    // ---- 8 .. 14 pairs: cv::findFundamentalMat switches to LMeDSPointSetRegistrator::run (`npoints >= 15` gate in fundam.cpp): same subsets from the same RNG, a model is
    //      ranked by the median of its errors (element count / 2 of the sorted errors), niters = max(RANSACUpdateNumIters(confidence, 0.45, 7, 1000), 3), no early exit
    const bool lmeds = count < 15;
    if (lmeds) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) { const int ni = sgx_ransac_update_num_iters(A.confidence, 0.45, 7, 1000); s_niters = ni > 3 ? ni : 3; s_minmed = DBL_MAX; }
        SGX_THREADS_END
        SGX_SYNC();
    }
    // ---- RANSACPointSetRegistrator::run, up to SGX_FM_CHUNK candidate groups per round (the first round draws only 8: on mostly static scenes the adaptive
    //      iteration count ends the loop after ~5-15 iterations, so scoring 32 x 3 models up front is wasted work)
    for (int round = 0;; round++) {
        const int G = round == 0 && !lmeds ? 8 : SGX_FM_CHUNK;
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {
            // cv::RNG::uniform(0, count) draws: seven distinct indices per group (getSubset's inner loops; the group is checked below)
            unsigned long long st = s_rng;
            for (int gI = 0; gI < G; gI++)
                for (int i = 0; i < 7; i++) {
                    int idx_i, j;
                    for (;;) {
                        st = (unsigned long long)(unsigned)st * 4164903690ull + (unsigned)(st >> 32);
                        idx_i = (int)((unsigned)st % (unsigned)count);
                        for (j = 0; j < i; j++) if (idx_i == g_idx[gI][j]) break;
                        if (j == i) break;
                    }
                    g_idx[gI][i] = idx_i;
                }
            s_rng = st;
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        if (tid < G) {
            float a[14], b[14];
            for (int i = 0; i < 7; i++) { const int id = g_idx[tid][i]; a[2 * i] = m1[2 * id]; a[2 * i + 1] = m1[2 * id + 1]; b[2 * i] = m2[2 * id]; b[2 * i + 1] = m2[2 * id + 1]; }
            int nm = -1;
            if (!sgx_fm_collinear7(a) && !sgx_fm_collinear7(b)) { nm = sgx_fm_run7point(a, b, g_work[tid], g_model[tid]); if (nm < 0 || nm > 3) nm = 0; }
            g_nmodels[tid] = nm;
            g_good[tid][0] = g_good[tid][1] = g_good[tid][2] = 0;
        }
        SGX_THREADS_END
        SGX_SYNC();
        if (lmeds) {
            // ---- the median error of every model of the round (one thread per model; at most 14 pairs)
            SGX_THREADS_BEGIN(tid)
            if (tid < 3 * G) {
                const int gI = tid / 3, k = tid - 3 * gI;
                if (k < g_nmodels[gI]) {
                    double Fm[9]; float err[16];
                    for (int i = 0; i < 9; i++) Fm[i] = g_model[gI][9 * k + i];
                    for (int p = 0; p < count; p++) {
                        const float v = sgx_fm_error(Fm, m1[2 * p], m1[2 * p + 1], m2[2 * p], m2[2 * p + 1]);
                        int b = p - 1; while (b >= 0 && err[b] > v) { err[b + 1] = err[b]; b--; }          /* insertion sort: nth_element(count / 2) leaves the same element there */
                        err[b + 1] = v;
                    }
                    g_med[gI][k] = err[count / 2];
                }
            }
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            if (tid == 0) {
                int iter = s_iter, attempts = s_attempts; const int niters = s_niters; double minmed = s_minmed;
                bool done = false;
                for (int gI = 0; gI < G && !done; gI++) {
                    if (iter >= niters) { done = true; break; }
                    if (g_nmodels[gI] < 0) { if (++attempts >= 10000) done = true; continue; }
                    attempts = 0;
                    for (int k = 0; k < g_nmodels[gI]; k++) {
                        const double med = (double)g_med[gI][k];
                        if (med < minmed) { minmed = med; for (int i = 0; i < 9; i++) s_best[i] = g_model[gI][9 * k + i]; s_best_iter = iter; s_best_root = k; }
                    }
                    iter++;
                }
                if (iter >= niters) done = true;
                s_iter = iter; s_attempts = attempts; s_minmed = minmed; s_done = done ? 1 : 0;
            }
            SGX_THREADS_END
            SGX_SYNC();
            if (s_done) break;
            continue;
        }
        // ---- findInliers for every model of the round
        for (int gI = 0; gI < G; gI++) {
            const int nm = g_nmodels[gI];
            for (int k = 0; k < nm; k++) {
                SGX_THREADS_BEGIN(tid)
                double Fm[9];
                for (int i = 0; i < 9; i++) Fm[i] = g_model[gI][9 * k + i];
                int cnt = 0;
                for (int p = tid; p < count; p += 256) cnt += sgx_fm_error(Fm, m1[2 * p], m1[2 * p + 1], m2[2 * p], m2[2 * p + 1]) <= t ? 1 : 0;
                if (cnt) sgx_atomic_add(&g_good[gI][k], cnt);
                SGX_THREADS_END
            }
        }
        SGX_SYNC();
        // ---- the sequential accept rule, replayed in iteration order (ptsetreg.cpp run())
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {
            int iter = s_iter, niters = s_niters, maxgood = s_maxgood, attempts = s_attempts;
            bool done = false;
            for (int gI = 0; gI < G && !done; gI++) {
                if (iter >= niters) { done = true; break; }
                if (g_nmodels[gI] < 0) {                                       /* getSubset repeats the draw (`continue` of its attempt loop), 10000 attempts at most */
                    if (++attempts >= 10000) done = true;                      /* getSubset fails: `return false` at iter 0 (nothing found yet), `break` later */
                    continue;
                }
                attempts = 0;
                for (int k = 0; k < g_nmodels[gI]; k++) {
                    const int good = g_good[gI][k];
                    if (good > (maxgood > 6 ? maxgood : 6)) {
                        for (int i = 0; i < 9; i++) s_best[i] = g_model[gI][9 * k + i];
                        maxgood = good; s_best_iter = iter; s_best_root = k;
                        niters = sgx_ransac_update_num_iters(A.confidence, (double)(count - good) / count, 7, niters);
                    }
                }
                iter++;
            }
            if (iter >= niters) done = true;
            s_iter = iter; s_niters = niters; s_maxgood = maxgood; s_attempts = attempts; s_done = done ? 1 : 0;
        }
        SGX_THREADS_END
        SGX_SYNC();
        if (s_done) break;
    }
    if (lmeds) {                                                     /* sigma from the best median, inliers of the best model, success = at least 7 of them */
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {
            int good = 0;
            if (s_minmed < DBL_MAX) {
                double sigma = 2.5 * 1.4826 * (1 + 5. / (count - 7)) * sqrt(s_minmed);
                if (sigma < 0.001) sigma = 0.001;
                const float ts = (float)(sigma * sigma);
                for (int p = 0; p < count; p++) good += sgx_fm_error(s_best, m1[2 * p], m1[2 * p + 1], m2[2 * p], m2[2 * p + 1]) <= ts ? 1 : 0;
            }
            s_maxgood = good >= 7 ? good : 0;
            if (A.stats) A.stats[4 * f + 3] = good;
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    SGX_THREADS_BEGIN(tid)
    const bool ok = s_maxgood > 0;
    if (tid < 9) A.F[9 * (size_t)f + tid] = ok ? s_best[tid] : 0.;
    if (tid == 0) {
        A.ok[f] = ok ? 1 : 0;
        if (A.stats) { A.stats[4 * f] = s_iter; A.stats[4 * f + 1] = s_best_iter; A.stats[4 * f + 2] = s_best_root; if (!lmeds) A.stats[4 * f + 3] = s_maxgood; }
    }
    SGX_THREADS_END
}
