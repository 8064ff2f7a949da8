// sgx_det_kernels.h — HIP kernels for the 2-D detector forward (MobileNetV3-SSDLite, fp32, NCHW planes like ncnn):
// pre-processing, pointwise convolution as an fp32-MFMA GEMM, depthwise / dense k x k convolution, elementwise ops,
// softmax, layout helpers.  Reference behaviour: src/sg-slam/src/Detector2D.cc:34-45 + the ncnn graph
// src/sg-slam/Thirdparty/ncnn_model/mobilenetv3_ssdlite_voc.param (layer semantics: oracle/detector_oracle.py).
#pragma once
#include "sgx_rt.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif

// XCD-aware work order.  Workgroup w of a launch (linear id, x fastest) is dispatched to XCD w % 8, and every XCD has its own 4 MB L2: with the plain order
// (work items of a frame on consecutive ids) neighbouring tiles of a frame land on eight different L2s and every halo / partial cache line they share is
// fetched from HBM once per XCD.  sgx_xcd_order deals the frames out to the XCDs instead: id -> (frame, item) such that all items of a frame run on the XCD
// frame % 8, consecutively (frames 8q .. 8q + 7 proceed in lock step, one per XCD).  A bijection of [0, per_frame * batch) for every batch: the frames past
// the last multiple of eight keep the plain order.  sgx_det_xcd_order = 0 restores the plain order everywhere (A/B switch, SGX_DET_XCD=0).
#ifndef SGX_EMU
__device__ int sgx_det_xcd_order = 1;
#else
static int sgx_det_xcd_order = 1;
#endif
SGX_DEV void sgx_xcd_order(int id, int per_frame, int batch, int *frame, int *item)
{
    const int b8 = batch & ~7;
    if (sgx_det_xcd_order && id < b8 * per_frame) {
        const int x = id & 7, i = id >> 3, q = i / per_frame;
        *item = i - q * per_frame; *frame = q * 8 + x;
    } else { const int f = id / per_frame; *frame = f; *item = id - f * per_frame; }
}

#define SGX_ACT_NONE 0
#define SGX_ACT_RELU 1
#define SGX_ACT_CLIP 2

SGX_DEV float sgx_act(float v, int act, float lo, float hi)
{
    if (act == SGX_ACT_RELU) return fmaxf(v, 0.f);
    if (act == SGX_ACT_CLIP) return fminf(fmaxf(v, lo), hi);
    return v;
}

// Epilogue program of a convolution: the chain of ncnn elementwise layers (BinaryOp / Clip / ReLU) that consumes its output, applied
// in registers in the graph's order with the graph's fp32 operations (so a fused plan is bit-identical to the unfused one).
// h-swish   x * clip(x + 3, 0, 6) / 6   = [ADD c] [CLIP] [MUL root] [DIV c];   SE gate + residual = [ADD c] [CLIP] [DIV c] [MUL t0] [ADD t1]
#define SGX_EPI_MAX 6
enum { SGX_EOP_ADD = 0, SGX_EOP_SUB, SGX_EOP_MUL, SGX_EOP_DIV, SGX_EOP_RSUB, SGX_EOP_RDIV, SGX_EOP_CLIP, SGX_EOP_RELU };
enum { SGX_ESRC_CONST = 0, SGX_ESRC_TENSOR, SGX_ESRC_ROOT };
struct SgxEpiStep { int op, src; float a, b; const float *t; };
// mode: the planner recognises the graph's recurring programs so the kernels run them as straight-line code (one uniform switch) instead
// of interpreting the step list; every mode performs exactly the steps' operations in the steps' order.
enum { SGX_EMODE_GENERIC = 0, SGX_EMODE_NONE, SGX_EMODE_ACT, SGX_EMODE_HSWISH, SGX_EMODE_GATE, SGX_EMODE_GATE_ADD, SGX_EMODE_ADD_T };
struct SgxEpi { int n; int mode; size_t tpitch; float c1, lo, hi, c2; const float *t0, *t1; SgxEpiStep s[SGX_EPI_MAX]; };   // tensor operands: output's shape, per-image pitch tpitch

// clip(v, lo, hi) = fminf(fmaxf(v, lo), hi) as ONE instruction.  The compiler turns the two calls into three (it canonicalises v in front of the max: IEEE mode) and cannot merge them
// into v_med3_f32 because lo / hi are run-time values; for lo <= hi the median IS the clip, including -0 against a +0 bound (the hardware orders -0 < +0 in max as in med3) and a NaN
// quiet-NaN operand (both return lo).  tools/ubench/div6_check.hip compares the two forms for all 2^32 operands with the graph's bounds (0, 6) and (0, +inf): they differ only for
// signalling NaNs, which no arithmetic instruction produces (an activation is always the result of one).
SGX_DEV float sgx_clipf(float v, float lo, float hi)
{
#ifndef SGX_EMU
    return __builtin_amdgcn_fmed3f(v, lo, hi);
#else
    return fminf(fmaxf(v, lo), hi);
#endif
}

// u / c for the divisor the graph uses everywhere (h-swish and h-sigmoid divide by 6): q0 = u r, e = fma(-q0, 6, u), q = fma(e, r, q0) with r = RN(1/6) is the correctly rounded quotient
// whenever none of the steps leaves the normal range; the sign is u's (a -0 numerator, which the network produces wherever x <= -3, would come out as +0).  Guard: t = u / 8 is
// subnormal, infinite or NaN  <=>  |u| < 2^-123 (non-zero), |u| = inf or NaN — then (the whole wave, so that no lane pays for both) the IEEE division runs.  Checked against u / 6.0f
// for all 2^32 operands (tools/check_div6.c: 0 differences).  Four instructions + two for the guard instead of the ten of the division sequence; 3.5 M activations per image.
SGX_DEV float sgx_div_c2(float u, float c2)
{
#ifndef SGX_EMU
    if (__builtin_bit_cast(unsigned, c2) == 0x40C00000u) {        // c2 == 6.0f, wave-uniform (a kernel argument); as an integer compare it runs on the scalar unit (there is no scalar float compare)
        const float t = u * 0.125f;
        unsigned long long bad;                                  // lanes whose t is sNaN | qNaN | -inf | -subnormal | +subnormal | +inf: the compare writes the lane mask straight into a scalar pair
        asm("v_cmp_class_f32_e64 %0, %1, %2" : "=s"(bad) : "v"(t), "v"(0x297));
        if (bad == 0) {
            const float r = 0x1.555556p-3f;                      // RN(1/6)
            const float q0 = u * r, e = fmaf(-q0, 6.0f, u), q = fmaf(e, r, q0);
            return __builtin_copysignf(q, u);
        }
    }
#endif
    return u / c2;
}

// tensor operand address = t + uoff (elements; wave-uniform in the tuned kernels -> scalar base register) + voff4 (bytes, 32-bit lane offset)
SGX_DEV float sgx_ldoff(const float *ubase, unsigned voff4) { return *(const float *)((const char *)ubase + voff4); }

template <int MODE>
SGX_DEV float sgx_epi_mode(const SgxEpi &e, float v, size_t uoff, unsigned voff4)
{
    if (MODE == SGX_EMODE_NONE) return v;
    if (MODE == SGX_EMODE_ACT) return sgx_clipf(v, e.lo, e.hi);                                                       // [RELU] (hi = +inf) / [CLIP]
    if (MODE == SGX_EMODE_HSWISH) { float u = v + e.c1; u = sgx_clipf(u, e.lo, e.hi); u = u * v; return sgx_div_c2(u, e.c2); }      // [ADD c][CLIP][MUL root][DIV c]
    if (MODE == SGX_EMODE_GATE) { float u = v + e.c1; u = sgx_clipf(u, e.lo, e.hi); u = sgx_div_c2(u, e.c2); return u * sgx_ldoff(e.t0 + uoff, voff4); }   // [ADD c][CLIP][DIV c][MUL t]
    if (MODE == SGX_EMODE_GATE_ADD) { float u = v + e.c1; u = sgx_clipf(u, e.lo, e.hi); u = sgx_div_c2(u, e.c2); u = u * sgx_ldoff(e.t0 + uoff, voff4); return u + sgx_ldoff(e.t1 + uoff, voff4); }
    if (MODE == SGX_EMODE_ADD_T) return v + sgx_ldoff(e.t1 + uoff, voff4);                                                // [ADD t]
    // generic interpreter, fully unrolled over the (at most SGX_EPI_MAX) steps: every field is a wave-uniform kernel argument at a
    // constant offset, so the scalar loads are hoisted out of the callers' loops
    const float root = v;
#pragma unroll
    for (int i = 0; i < SGX_EPI_MAX; i++) {
        if (i < e.n) {
            const int op = e.s[i].op;
            if (op == SGX_EOP_CLIP) v = fminf(fmaxf(v, e.s[i].a), e.s[i].b);
            else if (op == SGX_EOP_RELU) v = fmaxf(v, 0.f);
            else {
                const int src = e.s[i].src;
                const float o = src == SGX_ESRC_CONST ? e.s[i].a : (src == SGX_ESRC_ROOT ? root : sgx_ldoff(e.s[i].t + uoff, voff4));
                v = op == SGX_EOP_ADD ? v + o : op == SGX_EOP_MUL ? v * o : op == SGX_EOP_DIV ? v / o : op == SGX_EOP_SUB ? v - o : op == SGX_EOP_RSUB ? o - v : o / v;
            }
        }
    }
    return v;
}

SGX_DEV float sgx_epi(const SgxEpi &e, float v, size_t uoff, unsigned voff4 = 0)
{
    switch (e.mode) {
    case SGX_EMODE_NONE: return sgx_epi_mode<SGX_EMODE_NONE>(e, v, uoff, voff4);
    case SGX_EMODE_ACT: return sgx_epi_mode<SGX_EMODE_ACT>(e, v, uoff, voff4);
    case SGX_EMODE_HSWISH: return sgx_epi_mode<SGX_EMODE_HSWISH>(e, v, uoff, voff4);
    case SGX_EMODE_GATE: return sgx_epi_mode<SGX_EMODE_GATE>(e, v, uoff, voff4);
    case SGX_EMODE_GATE_ADD: return sgx_epi_mode<SGX_EMODE_GATE_ADD>(e, v, uoff, voff4);
    case SGX_EMODE_ADD_T: return sgx_epi_mode<SGX_EMODE_ADD_T>(e, v, uoff, voff4);
    default: return sgx_epi_mode<SGX_EMODE_GENERIC>(e, v, uoff, voff4);
    }
}

// ---------------------------------------------------------------------------------------------
// k_det_preprocess: ncnn::Mat::from_pixels_resize(PIXEL_RGB, w, h, 300, 300) + substract_mean_normalize (Detector2D.cc:39-40).
// ncnn resize_bilinear_c3: 11-bit fixed-point coefficients (host-built tables, clamp to (n-2, 1.0)), then u8 -> f32 - mean.
// out: [B][3][T][T]
// ---------------------------------------------------------------------------------------------
struct SgxDetTab { short o, a0, a1, pad; };

// One workgroup per (output row, image): the two source rows the row needs are staged in LDS as aligned dwords (coalesced), the T outputs x 3 channels
// are computed from LDS bytes and stored as three coalesced plane rows.  grid = (T, B); the row pitch and the image base are multiples of 4.
#define SGX_PRE_MAXW 2048
SGX_KERNEL(256) k_det_preprocess(int B, const uint8_t *img, int W, int H, int pitch, const SgxDetTab *xt, const SgxDetTab *yt, int T,
                                 float m0, float m1, float m2, float *out)
{
    SGX_LDS uint32_t rows[2][SGX_PRE_MAXW * 3 / 4 + 2];
    const int y = (int)blockIdx.x, b = (int)blockIdx.y;
    const SgxDetTab ty = yt[y];
    const int ndw = (3 * W + 3) >> 2;
    SGX_THREADS_BEGIN(tid)
    const uint32_t *s0 = (const uint32_t *)(img + ((size_t)b * H + ty.o) * pitch), *s1 = (const uint32_t *)(img + ((size_t)b * H + ty.o + 1) * pitch);
    for (int t = tid; t < ndw; t += 256) { rows[0][t] = s0[t]; rows[1][t] = s1[t]; }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const uint8_t *r0 = (const uint8_t *)rows[0], *r1 = (const uint8_t *)rows[1];
    const float mean[3] = { m0, m1, m2 };
    for (int x = tid; x < T; x += 256) {
        const SgxDetTab tx = xt[x];
        const uint8_t *p0 = r0 + 3 * tx.o, *p1 = r1 + 3 * tx.o;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int h0 = p0[c] * tx.a0 + p0[3 + c] * tx.a1, h1 = p1[c] * tx.a0 + p1[3 + c] * tx.a1;
            const int v = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
            out[(((size_t)b * 3 + c) * T + y) * T + x] = ((float)(v & 255) - mean[c]) * 1.0f;
        }
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_conv_pw: 1x1 convolution == GEMM  Out[oc][n] = sum_ic Wt[oc][ic] * In[ic][n] + bias[oc]   (n = pixel index, per image)
// on the fp32 matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TF peak, cdna guide §3).
// Workgroup = 4 waves = 64 (oc) x 64 (pixels) output tile, each wave one 32x32 accumulator (16 VGPR/lane); K staged
// through LDS in steps of 16 (8 MFMAs per wave per step).  A operand: lane l holds Wt[oc0 + (l&31)][k + (l>>5)],
// B operand: In[k + (l>>5)][n0 + (l&31)]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
// Epilogue: bias + the fused elementwise program, optional HWC store into a concat buffer (fuses ncnn Permute(3)+Flatten+Concat).
// grid = (ceil(N/64), ceil(outc/64), B)
// ---------------------------------------------------------------------------------------------
#define SGX_PW_KT 16
#ifndef SGX_EMU
typedef float sgx_f32x16 __attribute__((ext_vector_type(16)));
#endif

SGX_KERNEL(256) k_conv_pw(int inc, int outc, int N, const float *in, size_t in_pitch, const float *Wt, const float *bias,
                          float *out, size_t out_pitch, SgxEpi epi, int hwc, int hwc_off)
{
    SGX_LDS float As[SGX_PW_KT][64 + 1];      // [k][oc]
    SGX_LDS float Bs[SGX_PW_KT][64 + 1];      // [k][pixel]
    const int n0 = (int)blockIdx.x * 64, oc0 = (int)blockIdx.y * 64, b = (int)blockIdx.z;
    const float *X = in + (size_t)b * in_pitch;
    float *Y = out + (size_t)b * out_pitch;
#ifndef SGX_EMU
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    sgx_f32x16 acc;                                   // sum = bias, then the products in ascending k (ncnn's order)
#pragma unroll
    for (int r = 0; r < 16; r++) { const int row = oc0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); acc[r] = row < outc ? bias[row] : 0.f; }
    for (int k0 = 0; k0 < inc; k0 += SGX_PW_KT) {
        for (int t = tid; t < SGX_PW_KT * 64; t += 256) {
            const int kk = t >> 6, c = t & 63;                 // B tile: consecutive threads -> consecutive pixels (coalesced)
            const int k = k0 + kk;
            Bs[kk][c] = (k < inc && n0 + c < N) ? X[(size_t)k * N + n0 + c] : 0.f;
            const int ko = t & (SGX_PW_KT - 1), oc = t >> 4;   // A tile: consecutive threads -> consecutive k of one oc row
            As[ko][oc] = (k0 + ko < inc && oc0 + oc < outc) ? Wt[(size_t)(oc0 + oc) * inc + k0 + ko] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SGX_PW_KT; kk += 2) {
            const float a = As[kk + (lane >> 5)][wm + (lane & 31)];
            const float bb = Bs[kk + (lane >> 5)][wn + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = oc0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < outc && col < N) {
            const float v = sgx_epi(epi, acc[r], (size_t)b * epi.tpitch + (size_t)row * N + col);
            if (hwc) Y[(size_t)hwc_off + (size_t)col * outc + row] = v; else Y[(size_t)row * N + col] = v;
        }
    }
#else
    // kernel-logic emulator: same tile decomposition, scalar k-ordered FMA chain (what the fp32 MFMA computes)
    (void)As; (void)Bs;
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < 64 * 64; t += 256) {
        const int row = oc0 + (t >> 6), col = n0 + (t & 63);
        if (row < outc && col < N) {
            float s = bias[row];
            for (int k = 0; k < inc; k++) s = fmaf(Wt[(size_t)row * inc + k], X[(size_t)k * N + col], s);
            const float v = sgx_epi(epi, s, (size_t)b * epi.tpitch + (size_t)row * N + col);
            if (hwc) Y[(size_t)hwc_off + (size_t)col * outc + row] = v; else Y[(size_t)row * N + col] = v;
        }
    }
    SGX_THREADS_END
#endif
}

// ---------------------------------------------------------------------------------------------
// k_conv_kxk: k x k convolution, stride s, zero padding p; group == channels (depthwise, ncnn ConvolutionDepthWise) or
// group == 1 (dense, used by the 3x3 stride-2 stem).  One thread per output element; bandwidth-bound.
// grid = (ceil(Ho*Wo/256), outc, B)
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_conv_kxk(int inc, int outc, int H, int W, int Ho, int Wo, int k, int stride, int pad, int depthwise,
                           const float *in, size_t in_pitch, const float *Wt, const float *bias, float *out, size_t out_pitch, SgxEpi epi)
{
    SGX_THREADS_BEGIN(tid)
    const int idx = (int)blockIdx.x * 256 + tid, oc = (int)blockIdx.y, b = (int)blockIdx.z;
    if (idx < Ho * Wo) {
        const int oy = idx / Wo, ox = idx - oy * Wo;
        const float *X = in + (size_t)b * in_pitch;
        float s = bias[oc];
        const int c0 = depthwise ? oc : 0, c1 = depthwise ? oc + 1 : inc;
        for (int c = c0; c < c1; c++) {
            const float *w = Wt + ((size_t)oc * (depthwise ? 1 : inc) + (depthwise ? 0 : c)) * k * k;
            const float *xc = X + (size_t)c * H * W;
            for (int i = 0; i < k; i++) {
                const int iy = oy * stride - pad + i;
                if (iy < 0 || iy >= H) continue;
                for (int j = 0; j < k; j++) {
                    const int ix = ox * stride - pad + j;
                    if (ix < 0 || ix >= W) continue;
                    s = fmaf(w[i * k + j], xc[(size_t)iy * W + ix], s);
                }
            }
        }
        out[(size_t)b * out_pitch + (size_t)oc * Ho * Wo + idx] = sgx_epi(epi, s, (size_t)b * epi.tpitch + (size_t)oc * Ho * Wo + idx);
    }
    SGX_THREADS_END
}

// exact n / d for n < 2^20, d < 2^12 with m = ceil(2^32 / d) (host-computed): one mul-hi instead of an integer division
// (m == 0 encodes d == 1)
SGX_DEV unsigned sgx_fastdiv(unsigned n, unsigned m)
{
#ifndef SGX_EMU
    return m ? __umulhi(n, m) : n;
#else
    return m ? (unsigned)(((unsigned long long)n * m) >> 32) : n;
#endif
}

// ---------------------------------------------------------------------------------------------
// k_conv_pw2: 1x1 convolution as a weights-stationary streaming GEMM on the fp32 matrix cores.
//   Out[b][oc][n] = sum_k WtT[k][oc] * In[b][k][n] + bias[oc]          (WtT = weights transposed + zero-padded to [ceil32(inc)][ldw] on the host at load time)
// The pixel axis is flattened over the batch (g = b*N + n), so small feature maps (10x10 ... 1x1) still fill whole tiles.
// Workgroup = 4 waves; every wave owns PXB sub-tiles of 32 pixels and all OCB sub-tiles of 32 output channels of the block:
// its B operands (input) go global -> registers in the MFMA layout (lane l: In[k + (l>>5)][pixel l&31], two coalesced 128 B
// rows per load, prefetched 4 k-steps ahead) and are each used for OCB MFMAs; the A operands (weights) are staged through
// double-buffered LDS chunks of SGX_PW2_KC input channels, shared by the 4 waves, each used for PXB MFMAs.  Every input element
// is read once per oc block.  Rows of a chunk past `inc` hold zero weights, so a chunk is always processed in whole groups of 8 k.
// sum = bias, then the products in ascending k, as in k_conv_pw.  Epilogue: bias + fused elementwise program; CHW or HWC store.
// 1-D grid, XCD-aware: the oc blocks of one pixel tile run on the same XCD (shared L2) back to back.
// ---------------------------------------------------------------------------------------------
#define SGX_PW2_KC 32

#ifndef SGX_EMU
// read-back of one staged 32x32 tile, CHW store: lane = pixel (l31) of row 2j + half; scalar row base + 32-bit lane offsets
template <int MODE>
SGX_DEV void sgx_pw2_readback(const SgxEpi &epi, const float (*E)[33], int nj, int rt, int N, int half, int l31, float *out, unsigned ooff4, unsigned toff4)
{
#pragma unroll 4
    for (int j = 0; j < nj; j++) {
        const unsigned ubyte = (unsigned)(rt + 2 * j) * (unsigned)N * 4u;                             // wave-uniform row offset (blobs < 4 GB)
        *(float *)((char *)out + ubyte + ooff4) = sgx_epi_mode<MODE>(epi, E[2 * j + half][l31], (size_t)(ubyte >> 2), toff4);
    }
}

// CHW store straight from the accumulators: in the MFMA C layout a register already holds one output row segment per half-wave (lanes along pixels), so
// every store writes two whole 128 B lines; no LDS round trip, no barrier.  Row base wave-uniform, lane offset 32-bit.
template <int MODE, int OCB, int PXB>
SGX_DEV void sgx_pw2_store_direct(const SgxEpi &epi, const sgx_f32x16 (&acc)[OCB][PXB], int oc0, int outc, int N, int half, const bool (&pv)[PXB], float *out,
                                  const unsigned (&od)[PXB], const unsigned (&td)[PXB])
{
#pragma unroll
    for (int t = 0; t < OCB; t++) {
        if (oc0 + 32 * t < outc)                                         // uniform: oc padding of the last block
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rowu = oc0 + 32 * t + (r & 3) + 8 * (r >> 2);
            const unsigned ubyte = (unsigned)rowu * (unsigned)N * 4u;
            const bool rv = rowu + 4 * half < outc;
#pragma unroll
            for (int m = 0; m < PXB; m++)
                if (rv && pv[m]) *(float *)((char *)out + ubyte + od[m]) = sgx_epi_mode<MODE>(epi, acc[t][m][r], (size_t)(ubyte >> 2), td[m]);
        }
    }
}
#endif

#ifndef SGX_EMU
// Epilogue of the pointwise GEMM kernels (k_conv_pw2: exact fp32 MFMA; k_conv_pw3: bf16x3 MFMA — same accumulator layout): CHW outputs with a recognised program go
// straight from the accumulators; otherwise one 32x32 tile at a time through a wave-private LDS tile: a compact run-time loop applies the elementwise program
// (64 inlined copies of it cost hundreds of VGPRs), and the read-back order is chosen per store layout so stores stay contiguous (CHW: lanes along pixels; HWC:
// lanes along channels).
template <int OCB, int PXB>
SGX_DEV void sgx_pw2_epilogue(const SgxEpi &epi, const sgx_f32x16 (&acc)[OCB][PXB], float (*E)[33], int oc0, int outc, int N, int total, int g0, int half, int l31,
                              float *out, size_t out_pitch, const unsigned (&ooff4)[PXB], const unsigned (&toff4)[PXB], int hwc, int hwc_off, int direct)
{
    if (!hwc && direct && epi.mode != SGX_EMODE_GENERIC) {
        unsigned od[PXB], td[PXB]; bool pv[PXB];
#pragma unroll
        for (int m = 0; m < PXB; m++) {
            const unsigned h3 = 3u * (unsigned)half * (unsigned)N * 4u;     // ooff4 already carries half * N: rows of the upper half-wave are 4 further down
            od[m] = ooff4[m] + h3; td[m] = toff4[m] + h3; pv[m] = g0 + 32 * m + l31 < total;
        }
        switch (epi.mode) {
        case SGX_EMODE_NONE: sgx_pw2_store_direct<SGX_EMODE_NONE, OCB, PXB>(epi, acc, oc0, outc, N, half, pv, out, od, td); break;
        case SGX_EMODE_ACT: sgx_pw2_store_direct<SGX_EMODE_ACT, OCB, PXB>(epi, acc, oc0, outc, N, half, pv, out, od, td); break;
        case SGX_EMODE_HSWISH: sgx_pw2_store_direct<SGX_EMODE_HSWISH, OCB, PXB>(epi, acc, oc0, outc, N, half, pv, out, od, td); break;
        case SGX_EMODE_GATE: sgx_pw2_store_direct<SGX_EMODE_GATE, OCB, PXB>(epi, acc, oc0, outc, N, half, pv, out, od, td); break;
        case SGX_EMODE_GATE_ADD: sgx_pw2_store_direct<SGX_EMODE_GATE_ADD, OCB, PXB>(epi, acc, oc0, outc, N, half, pv, out, od, td); break;
        default: sgx_pw2_store_direct<SGX_EMODE_ADD_T, OCB, PXB>(epi, acc, oc0, outc, N, half, pv, out, od, td); break;
        }
        return;
    }
#pragma unroll 1
    for (int tile = 0; tile < OCB * PXB; tile++) {
        const int m = tile / OCB, t = tile - m * OCB;
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < OCB * PXB; tt++)
            if (tt == tile) {                                // static register indices, one uniform branch per tile
#pragma unroll
                for (int r = 0; r < 16; r++) E[(r & 3) + 8 * (r >> 2) + 4 * half][l31] = acc[tt % OCB][tt / OCB][r];
            }
        __syncthreads();
        unsigned oo = ooff4[0], to = toff4[0];
#pragma unroll
        for (int q = 1; q < PXB; q++) if (m == q) { oo = ooff4[q]; to = toff4[q]; }
        const int gt = g0 + 32 * m, rt = oc0 + 32 * t;
        if (!hwc) {
            const int nj = gt + l31 < total ? min(16, (outc - rt - half + 1) / 2) : 0;              // rows rt + half + 2j < outc
            switch (epi.mode) {
            case SGX_EMODE_NONE: sgx_pw2_readback<SGX_EMODE_NONE>(epi, E, nj, rt, N, half, l31, out, oo, to); break;
            case SGX_EMODE_ACT: sgx_pw2_readback<SGX_EMODE_ACT>(epi, E, nj, rt, N, half, l31, out, oo, to); break;
            case SGX_EMODE_HSWISH: sgx_pw2_readback<SGX_EMODE_HSWISH>(epi, E, nj, rt, N, half, l31, out, oo, to); break;
            case SGX_EMODE_GATE: sgx_pw2_readback<SGX_EMODE_GATE>(epi, E, nj, rt, N, half, l31, out, oo, to); break;
            case SGX_EMODE_GATE_ADD: sgx_pw2_readback<SGX_EMODE_GATE_ADD>(epi, E, nj, rt, N, half, l31, out, oo, to); break;
            case SGX_EMODE_ADD_T: sgx_pw2_readback<SGX_EMODE_ADD_T>(epi, E, nj, rt, N, half, l31, out, oo, to); break;
            default: sgx_pw2_readback<SGX_EMODE_GENERIC>(epi, E, nj, rt, N, half, l31, out, oo, to); break;
            }
        } else {
            const int row = rt + l31;
#pragma unroll 1
            for (int j = 0; j < 16; j++) {
                const int cc = 2 * j + half, g = gt + cc;
                const unsigned gg = (unsigned)min(g, total - 1), b = gg / (unsigned)N, n = gg - b * (unsigned)N;
                if (g < total && row < outc)
                    out[(size_t)b * out_pitch + (size_t)hwc_off + (size_t)n * outc + row] = sgx_epi(epi, E[l31][cc], (size_t)b * epi.tpitch + (size_t)row * N + n, 0);
            }
        }
    }
}
#endif

template <int OCB, int PXB>
SGX_KERNEL_OCC(256, (OCB * PXB <= 4 ? 4 : (OCB * PXB <= 6 ? 3 : 2))) k_conv_pw2(int inc, int outc, int N, int total, const float *in, size_t in_pitch, const float *WtT, const float *bias,
                           float *out, size_t out_pitch, SgxEpi epi, int hwc, int hwc_off, int nxt, int noc, int ldw, int direct)
{
    constexpr int OCT = 32 * OCB;
    SGX_LDS float Ws[2][SGX_PW2_KC][OCT + 1];       // weight chunks, double-buffered
    SGX_LDS float Es[4][32][33];                    // per-wave epilogue staging tile
    SGX_LDS float Bs[OCT];                          // bias of the oc block
    const int id = (int)blockIdx.x;
    const int grp = id / (8 * noc), rem = id - grp * (8 * noc);
    const int xt = grp * 8 + (rem & 7), yt = rem >> 3;                  // XCD = id % 8 = xt % 8
    if (xt >= nxt) return;                                               // uniform per workgroup
    const int oc0 = yt * OCT;
#ifndef SGX_EMU
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int g0 = (xt * 4 + wave) * (32 * PXB);
    // Addressing: every global access is (wave-uniform base: channel row / weight row) + (32-bit per-lane byte offset: image, pixel, k parity),
    // i.e. scalar-base + vector-offset loads/stores with no per-access address arithmetic (host guarantees blobs < 4 GB and an even `inc`).
    // Lanes past the end of the pixel axis (and rows past outc, k past inc) work on clamped addresses: finite values that meet zero
    // weights or are never stored.  No divergent control flow anywhere in the main loop.
    unsigned ioff4[PXB], ooff4[PXB], toff4[PXB];
#pragma unroll
    for (int m = 0; m < PXB; m++) {
        const unsigned gg = (unsigned)min(g0 + 32 * m + l31, total - 1), b = gg / (unsigned)N, n = gg - b * (unsigned)N, hn = n + (unsigned)half * (unsigned)N;
        ioff4[m] = (b * (unsigned)in_pitch + hn) * 4u; ooff4[m] = (b * (unsigned)out_pitch + hn) * 4u; toff4[m] = (b * (unsigned)epi.tpitch + hn) * 4u;
    }
    constexpr int D = OCB * PXB >= 4 ? 4 : (OCB * PXB == 2 ? 8 : 16);   // B-operand prefetch ring: D k-steps (of 2 input channels) ahead of the MFMAs, ~1000+ MFMA cycles
    constexpr int WR = SGX_PW2_KC * OCT / 256;             // weight-chunk elements per thread
    float bq[D][PXB];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float *rowp = (const float *)((const char *)in + (unsigned)min(2 * d, inc - 2) * (unsigned)N * 4u);
#pragma unroll
        for (int m = 0; m < PXB; m++) bq[d][m] = sgx_ldoff(rowp, ioff4[m]);
    }
    // weights: host-padded with zeros to [ceil32(inc)][ldw] (ldw >= any oc block end), so chunk loads are unconditional: uniform base + lane offset
    float wr[WR]; unsigned woff4[WR];
#pragma unroll
    for (int i = 0; i < WR; i++) { const int t = tid + 256 * i, kk = t / OCT, oc = t - kk * OCT; woff4[i] = (unsigned)(kk * ldw + oc) * 4u; }
#pragma unroll
    for (int i = 0; i < WR; i++) wr[i] = sgx_ldoff(WtT + oc0, woff4[i]);
#pragma unroll
    for (int i = 0; i < WR; i++) { const int t = tid + 256 * i, kk = t / OCT, oc = t - kk * OCT; Ws[0][kk][oc] = wr[i]; }
    if (tid < OCT) Bs[tid] = bias[min(oc0 + tid, outc - 1)];
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
    int buf = 0;
    for (int k0 = 0; k0 < inc; k0 += SGX_PW2_KC, buf ^= 1) {
        const bool more = k0 + SGX_PW2_KC < inc;
        if (more) {                                        // next weight chunk: global -> registers now, -> LDS after this chunk's MFMAs
            const float *wb = WtT + (size_t)(k0 + SGX_PW2_KC) * ldw + oc0;
#pragma unroll
            for (int i = 0; i < WR; i++) wr[i] = sgx_ldoff(wb, woff4[i]);
        }
        const int kend = min(SGX_PW2_KC, inc - k0);
        for (int kk = 0; kk < kend; kk += 2 * D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                float bv[PXB];
                const float *rowp = (const float *)((const char *)in + (unsigned)min(k0 + kk + 2 * (d + D), inc - 2) * (unsigned)N * 4u);   // wave-uniform row base
#pragma unroll
                for (int m = 0; m < PXB; m++) { bv[m] = bq[d][m]; bq[d][m] = sgx_ldoff(rowp, ioff4[m]); }
                if (kk + 2 * d < kend) {                     // (uniform) rows past the end of the chunk hold zero weights: skip their MFMAs
#pragma unroll
                    for (int t = 0; t < OCB; t++) {
                        const float a = Ws[buf][kk + 2 * d + half][32 * t + l31];
#pragma unroll
                        for (int m = 0; m < PXB; m++) acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[m], acc[t][m], 0, 0, 0);
                    }
                }
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < WR; i++) { const int t = tid + 256 * i, kk = t / OCT, oc = t - kk * OCT; Ws[buf ^ 1][kk][oc] = wr[i]; }
        }
        __syncthreads();
    }
    sgx_pw2_epilogue<OCB, PXB>(epi, acc, Es[wave], oc0, outc, N, total, g0, half, l31, out, out_pitch, ooff4, toff4, hwc, hwc_off, direct);
#else
    (void)Ws; (void)Es; (void)Bs;
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < OCT * 128 * PXB; t += 256) {
        const int row = oc0 + t / (128 * PXB), g = xt * 128 * PXB + t % (128 * PXB);
        if (row < outc && g < total) {
            const int b = g / N, n = g - b * N;
            const float *X = in + (size_t)b * in_pitch;
            float s = bias[row];
            for (int k = 0; k < inc; k++) s = fmaf(WtT[(size_t)k * ldw + row], X[(size_t)k * N + n], s);
            const float v = sgx_epi(epi, s, (size_t)b * epi.tpitch + (size_t)row * N + n);
            float *Y = out + (size_t)b * out_pitch;
            if (hwc) Y[(size_t)hwc_off + (size_t)n * outc + row] = v; else Y[(size_t)row * N + n] = v;
        }
    }
    SGX_THREADS_END
#endif
}

// ---------------------------------------------------------------------------------------------
// k_conv_dw: depthwise K x K convolution (ncnn ConvolutionDepthWise, group == channels), stride s, zero padding.
// Planes are indexed p = b*C + c over the dense [B][C][H][W] blob.  A workgroup owns P consecutive planes x one band of RB output
// rows: the zero-padded input band is staged in LDS with coalesced row loads (each input element read once), every thread then
// produces outputs o = tid, tid+256, ... of the band (consecutive threads -> consecutive x: conflict-free LDS reads at stride 1,
// coalesced stores).  Small planes (19x19 ... 1x1) are grouped P per workgroup, large ones (150x150) are cut into row bands.
// Tap order (i, j) ascending with fmaf, as k_conv_kxk; taps that fall into the padding add an exact zero.
// grid = (ceil(B*C / P) * nbands)
// ---------------------------------------------------------------------------------------------
template <int K>
SGX_KERNEL(256) k_conv_dw(int C, int H, int W, int Ho, int Wo, int stride, int pad, int P, int RB, int nbands, int nplanes,
                          unsigned per_magic, unsigned wp_magic, unsigned wo_magic, const float *in, const float *Wt, const float *bias, float *out, SgxEpi epi)
{
    SGX_DYN_LDS(smem);
    float *tile = (float *)smem;
    const int grp = (int)blockIdx.x / nbands, band = (int)blockIdx.x - grp * nbands;
    const int p0 = grp * P, np = min(P, nplanes - p0);
    const int r0 = band * RB, nrows = min(RB, Ho - r0);
    const int Wp = (Wo - 1) * stride + K, Rin = (nrows - 1) * stride + K, iy0 = r0 * stride - pad;
    float *wl = tile + (size_t)P * ((RB - 1) * stride + K) * Wp;          // [P][K*K] weights + [P] bias
    SGX_THREADS_BEGIN(tid)
    // staging: linear index over the P padded bands, 8 independent loads in flight per thread before the LDS stores (latency, not
    // instruction count, bounds this phase); consecutive threads -> consecutive x: coalesced row segments
    const int per = Rin * Wp, tot = np * per;
    for (int t0 = tid; t0 < tot; t0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = min(t0 + 256 * u, tot - 1);
            const int q = P == 1 ? 0 : (int)sgx_fastdiv((unsigned)t, per_magic), r = t - q * per;
            const int ry = (int)sgx_fastdiv((unsigned)r, wp_magic), cx = r - ry * Wp;
            const int iy = iy0 + ry, ix = cx - pad;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            v[u] = in[((size_t)(p0 + q) * H + (ok ? iy : 0)) * W + (ok ? ix : 0)];
            v[u] = ok ? v[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int t = t0 + 256 * u; if (t < tot) tile[t] = v[u]; }
    }
    for (int t = tid; t < np * (K * K + 1); t += 256) {
        const int q = t / (K * K + 1), j = t - q * (K * K + 1), c = (p0 + q) % C;
        wl[t] = j < K * K ? Wt[(size_t)c * K * K + j] : bias[c];
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int per = Rin * Wp, nout = nrows * Wo;
    for (int q = 0; q < np; q++) {
        const float *w = wl + q * (K * K + 1);
        float wr[K * K];
#pragma unroll
        for (int j = 0; j < K * K; j++) wr[j] = w[j];
        const float bz = w[K * K];
        const size_t obase = (size_t)(p0 + q) * Ho * Wo + (size_t)r0 * Wo;
        for (int o = tid; o < nout; o += 256) {
            const int oy = (int)sgx_fastdiv((unsigned)o, wo_magic), ox = o - oy * Wo;
            const float *base = tile + q * per + (oy * stride) * Wp + ox * stride;
            float s = bz;
#pragma unroll
            for (int i = 0; i < K; i++)
#pragma unroll
                for (int j = 0; j < K; j++) s = fmaf(wr[i * K + j], base[i * Wp + j], s);
            out[obase + o] = sgx_epi(epi, s, obase + o);
        }
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_conv_dw2<K, S>: depthwise K x K convolution, stride S — the lean version of k_conv_dw (same decomposition: a workgroup owns P consecutive planes x one band of
// RB output rows, the zero-padded input band staged once in LDS), with both phases register-blocked 4 wide:
//   staging  one group of 4 consecutive padded columns per thread and step (4 scalar global loads, ONE ds_write_b128); the (plane, row, group) counters advance
//            incrementally — no division per element;
//   compute  one task = 4 consecutive outputs of a row: the (3 S + K) inputs of each of the K rows come in as aligned ds_read_b128 (LDS row pitch a multiple of 4, block origin
//            4 bx S), weights + bias of the plane as K*K + 1 floats padded to whole float4s; 4 K^2 fmaf in tap order (i, j) ascending per output, as k_conv_kxk; the
//            epilogue program is selected once per kernel, not per output; a full block leaves as one 16-byte store (consecutive lanes -> consecutive 16 B).
// LDS: [P][Rmax][pitch] floats + [P][KW] weights, Rmax = (RB - 1) S + K, pitch = roundup4((ceil(Wo / 4) - 1) 4 S + 3 S + K), KW = roundup4(K K + 1).
// grid = ceil(B C / P) * nbands.
// ---------------------------------------------------------------------------------------------
struct alignas(16) sgx_f4 { float v[4]; };

#ifndef SGX_STAGE_U
#define SGX_STAGE_U 4
#endif
// LDS staging shared by k_conv_dw2 / k_conv_stem2: np planes (plane q at in + q * H * W), rows iy0 .. iy0 + Rin - 1, padded columns 0 .. pitch - 1 (column cx = input
// column cx - pad), zeros outside the image; tile[q][ry][cx] with plane stride plane_stride.  One group of 4 columns per thread and step (4 scalar global loads, one
// ds_write_b128); the group index g = tid + 256 k runs over (q, ry, column group cg), cg fastest, and the counters advance by the constant step 256 = (dq, dry, dcg).
SGX_DEV void sgx_stage_planes4(int tid, const float *in, int np, int H, int W, int iy0, int Rin, int pad, int pitch, int plane_stride, float *tile)
{
    const int G = pitch >> 2;
    const int rows256 = 256 / G, dcg = 256 - rows256 * G, dq = rows256 / Rin, dry = rows256 - dq * Rin;
    const int ngrp = np * Rin * G;
    int ry = tid / G, cg = tid - ry * G, q = ry / Rin; ry -= q * Rin;
    for (int g0 = tid; g0 < ngrp; g0 += 256 * SGX_STAGE_U) {
        float v[SGX_STAGE_U][4]; int dst[SGX_STAGE_U];
#pragma unroll
        for (int u = 0; u < SGX_STAGE_U; u++) {                     // 4 SGX_STAGE_U independent loads in flight per thread before the LDS stores
            const int qq = min(q, np - 1), iy = iy0 + ry;
            const bool rowok = (unsigned)iy < (unsigned)H;
            const float *src = in + ((size_t)qq * H + (size_t)(rowok ? iy : 0)) * W;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int ix = 4 * cg + x - pad;
                const bool ok = rowok && (unsigned)ix < (unsigned)W;
                const float ld = src[ok ? ix : 0];
                v[u][x] = ok ? ld : 0.f;
            }
            dst[u] = g0 + 256 * u < ngrp ? qq * plane_stride + ry * pitch + 4 * cg : -1;
            cg += dcg; const int carry = cg >= G ? 1 : 0; cg -= carry ? G : 0;
            ry += dry + carry; q += dq; if (ry >= Rin) { ry -= Rin; q++; }
        }
#pragma unroll
        for (int u = 0; u < SGX_STAGE_U; u++)
            if (dst[u] >= 0) { sgx_f4 pk; pk.v[0] = v[u][0]; pk.v[1] = v[u][1]; pk.v[2] = v[u][2]; pk.v[3] = v[u][3]; *(sgx_f4 *)(tile + dst[u]) = pk; }
    }
}

template <int K, int S, int MODE>
SGX_DEV void sgx_dw2_tasks(int tid, int np, int nrows, int nbx, int Wo, int Ho, int r0, int p0, int pitch, int plane_stride, unsigned task_magic, unsigned nbx_magic,
                           const float *tile, const float *wl, float *out, const SgxEpi &epi)
{
    constexpr int KW = (K * K + 1 + 3) & ~3, NV = (3 * S + K + 3) / 4;
    const int per_plane = nrows * nbx, ntask = np * per_plane;
    for (int t = tid; t < ntask; t += 256) {
        const int q = np == 1 ? 0 : (int)sgx_fastdiv((unsigned)t, task_magic), rem = t - q * per_plane;
        const int oy = (int)sgx_fastdiv((unsigned)rem, nbx_magic), bx = rem - oy * nbx;
        float w[KW];
        const sgx_f4 *wq = (const sgx_f4 *)(wl + q * KW);
#pragma unroll
        for (int g = 0; g < KW / 4; g++) { const sgx_f4 x = wq[g]; w[4 * g] = x.v[0]; w[4 * g + 1] = x.v[1]; w[4 * g + 2] = x.v[2]; w[4 * g + 3] = x.v[3]; }
        float acc[4] = { w[K * K], w[K * K], w[K * K], w[K * K] };
        const float *base = tile + q * plane_stride + (oy * S) * pitch + bx * (4 * S);
#pragma unroll
        for (int i = 0; i < K; i++) {
            float v[4 * NV];
            const sgx_f4 *row = (const sgx_f4 *)(base + i * pitch);
#pragma unroll
            for (int g = 0; g < NV; g++) { const sgx_f4 x = row[g]; v[4 * g] = x.v[0]; v[4 * g + 1] = x.v[1]; v[4 * g + 2] = x.v[2]; v[4 * g + 3] = x.v[3]; }
#pragma unroll
            for (int j = 0; j < K; j++)
#pragma unroll
                for (int x = 0; x < 4; x++) acc[x] = fmaf(w[i * K + j], v[x * S + j], acc[x]);
        }
        const size_t o = ((size_t)(p0 + q) * Ho + (size_t)(r0 + oy)) * Wo + (size_t)(4 * bx);
        const int nx = min(4, Wo - 4 * bx);
        float res[4];
#pragma unroll
        for (int x = 0; x < 4; x++) res[x] = sgx_epi_mode<MODE>(epi, acc[x], o + (size_t)min(x, nx - 1), 0);
        if (nx == 4) { sgx_f4 pk; pk.v[0] = res[0]; pk.v[1] = res[1]; pk.v[2] = res[2]; pk.v[3] = res[3]; memcpy(out + o, &pk, 16); }
        else for (int x = 0; x < nx; x++) out[o + x] = res[x];
    }
}

template <int K, int S>
SGX_KERNEL(256) k_conv_dw2(int C, int H, int W, int Ho, int Wo, int pad, int P, int RB, int nbands, int nplanes, int pitch, unsigned task_magic, unsigned nbx_magic,
                           const float *in, const float *Wt, const float *bias, float *out, SgxEpi epi)
{
    SGX_DYN_LDS(smem);
    constexpr int KW = (K * K + 1 + 3) & ~3;
    float *tile = (float *)smem;
    const int grp = (int)blockIdx.x / nbands, band = (int)blockIdx.x - grp * nbands;
    const int p0 = grp * P, np = min(P, nplanes - p0);
    const int r0 = band * RB, nrows = min(RB, Ho - r0);
    const int Rmax = (RB - 1) * S + K, Rin = (nrows - 1) * S + K, iy0 = r0 * S - pad;
    const int plane_stride = Rmax * pitch;
    float *wl = tile + (size_t)P * plane_stride;
    SGX_THREADS_BEGIN(tid)
    sgx_stage_planes4(tid, in + (size_t)p0 * H * W, np, H, W, iy0, Rin, pad, pitch, plane_stride, tile);
    for (int t = tid; t < np * KW; t += 256) {
        const int qw = t / KW, j = t - qw * KW, c = (p0 + qw) % C;
        wl[t] = j < K * K ? Wt[(size_t)c * K * K + j] : (j == K * K ? bias[c] : 0.f);
    }
    SGX_THREADS_END
    SGX_SYNC();
    const int nbx = (Wo + 3) >> 2;
    SGX_THREADS_BEGIN(tid)
#define SGX_DW2_RUN(M) sgx_dw2_tasks<K, S, M>(tid, np, nrows, nbx, Wo, Ho, r0, p0, pitch, plane_stride, task_magic, nbx_magic, tile, wl, out, epi)
    switch (epi.mode) {
    case SGX_EMODE_NONE: SGX_DW2_RUN(SGX_EMODE_NONE); break;
    case SGX_EMODE_ACT: SGX_DW2_RUN(SGX_EMODE_ACT); break;
    case SGX_EMODE_HSWISH: SGX_DW2_RUN(SGX_EMODE_HSWISH); break;
    case SGX_EMODE_GATE: SGX_DW2_RUN(SGX_EMODE_GATE); break;
    case SGX_EMODE_GATE_ADD: SGX_DW2_RUN(SGX_EMODE_GATE_ADD); break;
    case SGX_EMODE_ADD_T: SGX_DW2_RUN(SGX_EMODE_ADD_T); break;
    default: SGX_DW2_RUN(SGX_EMODE_GENERIC); break;
    }
#undef SGX_DW2_RUN
    SGX_THREADS_END
}

// Two fp32 lanes of one v_pk_fma_f32: a wave64 v_fma_f32 occupies the SIMD for 4 cycles, the packed form does two FMAs in the same slot (that is how gfx950 reaches its
// 157 TFLOP/s fp32 vector peak).  Each half is an ordinary fused multiply-add, so results equal the scalar chain bit for bit.  The weights stay in SCALAR registers:
// gfx950 reads an SGPR pair as a packed source with full pair and op_sel semantics (checked on hardware, tools/ubench/pk_fma_sgpr.hip); the compiler never emits that form
// (it copies scalars into VGPRs first, one v_mov per use), hence the inline assembly.
#ifndef SGX_EMU
typedef float sgx_f2 __attribute__((ext_vector_type(2)));
SGX_DEV sgx_f2 sgx_mk2(float a, float b) { sgx_f2 r; r.x = a; r.y = b; return r; }
// c + (w.x, w.x) * b   and   c + (w.y, w.y) * b, w in scalar registers
SGX_DEV sgx_f2 sgx_fma2_wlo(sgx_f2 w, sgx_f2 b, sgx_f2 c) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(c) : "s"(w), "v"(b)); return c; }
SGX_DEV sgx_f2 sgx_fma2_whi(sgx_f2 w, sgx_f2 b, sgx_f2 c) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c) : "s"(w), "v"(b)); return c; }
// c + w * b, w a scalar-register pair
SGX_DEV sgx_f2 sgx_fma2_w(sgx_f2 w, sgx_f2 b, sgx_f2 c) { asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "s"(w), "v"(b)); return c; }
// c + w * (d.x, d.x)   and   c + w * (d.y, d.y), w a scalar-register pair
SGX_DEV sgx_f2 sgx_fma2_w_dlo(sgx_f2 w, sgx_f2 d, sgx_f2 c) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(c) : "s"(w), "v"(d)); return c; }
SGX_DEV sgx_f2 sgx_fma2_w_dhi(sgx_f2 w, sgx_f2 d, sgx_f2 c) { asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(c) : "s"(w), "v"(d)); return c; }
#else
struct sgx_f2 { float x, y; };
static inline sgx_f2 sgx_mk2(float a, float b) { sgx_f2 r; r.x = a; r.y = b; return r; }
static inline sgx_f2 sgx_fma2_w(sgx_f2 a, sgx_f2 b, sgx_f2 c) { sgx_f2 r; r.x = fmaf(a.x, b.x, c.x); r.y = fmaf(a.y, b.y, c.y); return r; }
static inline sgx_f2 sgx_fma2_wlo(sgx_f2 w, sgx_f2 b, sgx_f2 c) { return sgx_fma2_w(sgx_mk2(w.x, w.x), b, c); }
static inline sgx_f2 sgx_fma2_whi(sgx_f2 w, sgx_f2 b, sgx_f2 c) { return sgx_fma2_w(sgx_mk2(w.y, w.y), b, c); }
static inline sgx_f2 sgx_fma2_w_dlo(sgx_f2 w, sgx_f2 d, sgx_f2 c) { return sgx_fma2_w(w, sgx_mk2(d.x, d.x), c); }
static inline sgx_f2 sgx_fma2_w_dhi(sgx_f2 w, sgx_f2 d, sgx_f2 c) { return sgx_fma2_w(w, sgx_mk2(d.y, d.y), c); }
#endif

// ---------------------------------------------------------------------------------------------
// k_conv_stem2<INC>: the 3 x 3, stride-2 stem (INC input channels -> up to 16 output channels), lean version of k_conv_stem: a workgroup owns a band of RB output rows of
// one image, the INC padded input bands staged by sgx_stage_planes4; one task = 4 consecutive output pixels x ALL output channels (64 accumulators): per (c, i) row three
// aligned ds_read_b128 bring the 9 inputs, the 16 weights of a tap are wave-uniform (scalar loads from the host-transposed table WtT[(c*9 + i*3 + j)][16]) and feed 64 fmaf.
// Accumulation order per output: c, i, j ascending (as k_conv_kxk).  Every channel's 4 results leave as one 16-byte store.  grid = (nbands, B)
// ---------------------------------------------------------------------------------------------
template <int INC, int MODE>
SGX_DEV void sgx_stem2_tasks(int tid, int outc, int nrows, int nbx, int Wo, int Ho, int r0, int pitch, int plane_stride, unsigned nbx_magic,
                             const float *tile, const float *__restrict__ WtT, const float *__restrict__ bias, float *__restrict__ out, size_t tbase, const SgxEpi &epi)
{
    const int ntask = nrows * nbx;
    for (int t = tid; t < ntask; t += 256) {
        const int oy = (int)sgx_fastdiv((unsigned)t, nbx_magic), bx = t - oy * nbx;
        int zoff = 0;
#ifndef SGX_EMU
        asm volatile("" : "+s"(zoff));                            // opaque (but scalar) zero: keeps the 27 x 16 weight loads inside the task loop as scalar loads instead of 432 hoisted VGPRs
#endif
        // Round 6: the 16 output channels as eight PAIRS per pixel — one v_pk_fma_f32 per pair and tap with the two weights as a scalar-register pair and the pixel's value
        // broadcast from its half of a vector-register pair (sgx_fma2_w_dlo / _dhi: the forms of the detector's depthwise taps) — 216 instead of 432 multiply-add instructions
        // per output pixel; per output the same fmaf chain in the same order, so the same bits.
        sgx_f2 acc[8][4];
#pragma unroll
        for (int op = 0; op < 8; op++) { const sgx_f2 bz = sgx_mk2(bias[min(2 * op, outc - 1)], bias[min(2 * op + 1, outc - 1)]); acc[op][0] = bz; acc[op][1] = bz; acc[op][2] = bz; acc[op][3] = bz; }
#pragma unroll 1
        for (int c = 0; c < INC; c++)                               // run-time loops over (c, i): 3 taps x 16 scalar weights live at a time
#pragma unroll 1
            for (int i = 0; i < 3; i++) {
                sgx_f2 v[6];                                           // twelve consecutive input columns as six register pairs
                const sgx_f4 *row = (const sgx_f4 *)(tile + c * plane_stride + (2 * oy + i) * pitch + 8 * bx);
#pragma unroll
                for (int g = 0; g < 3; g++) { const sgx_f4 x = row[g]; v[2 * g] = sgx_mk2(x.v[0], x.v[1]); v[2 * g + 1] = sgx_mk2(x.v[2], x.v[3]); }
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float *w = WtT + zoff + ((c * 3 + i) * 3 + j) * 16;
#pragma unroll
                    for (int op = 0; op < 8; op++) {
                        const sgx_f2 wp = sgx_mk2(w[2 * op], w[2 * op + 1]);
#pragma unroll
                        for (int x = 0; x < 4; x++) {                       // input column 2 x + j: the low or the high half of pair (2 x + j) / 2
                            if (((2 * x + j) & 1) == 0) acc[op][x] = sgx_fma2_w_dlo(wp, v[(2 * x + j) >> 1], acc[op][x]);
                            else acc[op][x] = sgx_fma2_w_dhi(wp, v[(2 * x + j) >> 1], acc[op][x]);
                        }
                    }
                }
            }
        const size_t pix = (size_t)(r0 + oy) * Wo + (size_t)(4 * bx);
        const int nx = min(4, Wo - 4 * bx);
#pragma unroll
        for (int oc = 0; oc < 16; oc++)
            if (oc < outc) {
                const size_t idx = (size_t)oc * Ho * Wo + pix;
                float res[4];
#pragma unroll
                for (int x = 0; x < 4; x++) res[x] = sgx_epi_mode<MODE>(epi, (oc & 1) ? acc[oc >> 1][x].y : acc[oc >> 1][x].x, tbase + idx + (size_t)min(x, nx - 1), 0);
                if (nx == 4) { sgx_f4 pk; pk.v[0] = res[0]; pk.v[1] = res[1]; pk.v[2] = res[2]; pk.v[3] = res[3]; memcpy(out + idx, &pk, 16); }
                else for (int x = 0; x < nx; x++) out[idx + x] = res[x];
            }
    }
}

template <int INC>
SGX_KERNEL(256) k_conv_stem2(int outc, int H, int W, int Ho, int Wo, int pad, int RB, int pitch, unsigned nbx_magic,
                             const float *__restrict__ in, size_t in_pitch, const float *__restrict__ WtT, const float *__restrict__ bias, float *__restrict__ out, size_t out_pitch, SgxEpi epi)
{
    SGX_DYN_LDS(smem);
    float *tile = (float *)smem;
    const int b = (int)blockIdx.y, r0 = (int)blockIdx.x * RB, nrows = min(RB, Ho - r0);
    const int Rmax = (RB - 1) * 2 + 3, Rin = (nrows - 1) * 2 + 3, iy0 = r0 * 2 - pad, plane_stride = Rmax * pitch;
    SGX_THREADS_BEGIN(tid)
    sgx_stage_planes4(tid, in + (size_t)b * in_pitch, INC, H, W, iy0, Rin, pad, pitch, plane_stride, tile);
    SGX_THREADS_END
    SGX_SYNC();
    const int nbx = (Wo + 3) >> 2;
    SGX_THREADS_BEGIN(tid)
#define SGX_STEM2_RUN(M) sgx_stem2_tasks<INC, M>(tid, outc, nrows, nbx, Wo, Ho, r0, pitch, plane_stride, nbx_magic, tile, WtT, bias, out + (size_t)b * out_pitch, (size_t)b * epi.tpitch, epi)
    switch (epi.mode) {
    case SGX_EMODE_NONE: SGX_STEM2_RUN(SGX_EMODE_NONE); break;
    case SGX_EMODE_ACT: SGX_STEM2_RUN(SGX_EMODE_ACT); break;
    default: SGX_STEM2_RUN(SGX_EMODE_HSWISH); break;             // the host launches this kernel for these three programs only
    }
#undef SGX_STEM2_RUN
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_stem_pre: k_det_preprocess + k_conv_stem2<3> in one kernel — the 3 x T x T fp32 network input never exists in HBM (1.08 MB written and read back per frame
// otherwise).  A workgroup owns a band of RB output rows of one image, as k_conv_stem2; instead of copying the three padded input planes of the band from HBM it
// COMPUTES them into the same LDS tile from the interleaved u8 image: every tile element (row, column) is one resized pixel — the integer bilinear resize of
// k_det_preprocess (same tables, same arithmetic) on 6 + 6 source bytes read straight from the image (neighbouring lanes read neighbouring bytes: the rows stay in
// L1 / L2), minus the channel mean — or the convolution's zero padding.  The rows of the band's halo (2 of 2 RB + 1) are resized twice; everything after the barrier
// is k_conv_stem2.  grid = nbands * B in the XCD-aware order.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_stem_pre(const uint8_t *__restrict__ img, int H, int ipitch, const SgxDetTab *__restrict__ xt, const SgxDetTab *__restrict__ yt, int T, float m0, float m1, float m2,
                           int nbands, int batch, int outc, int Ho, int Wo, int pad, int RB, int pitch, unsigned pitch_magic, unsigned nbx_magic,
                           const float *__restrict__ WtT, const float *__restrict__ bias, float *__restrict__ out, size_t out_pitch, SgxEpi epi)
{
    SGX_DYN_LDS(smem);
    float *tile = (float *)smem;
    int b, band;
    sgx_xcd_order((int)blockIdx.x, nbands, batch, &b, &band);
    const int r0 = band * RB, nrows = min(RB, Ho - r0);
    const int Rmax = (RB - 1) * 2 + 3, Rin = (nrows - 1) * 2 + 3, iy0 = r0 * 2 - pad, plane_stride = Rmax * pitch;
    // the resize tables of the band in LDS (all T columns, the band's rows): a tile element then needs ONE round of global loads (its 12 source bytes)
    SgxDetTab *xs = (SgxDetTab *)(tile + 3 * plane_stride), *ys = xs + T;
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < T; i += 256) xs[i] = xt[i];
    if (tid < Rin) ys[tid] = yt[min(max(iy0 + tid, 0), T - 1)];
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const uint8_t *base = img + (size_t)b * H * ipitch;
    const float mean[3] = { m0, m1, m2 };
#ifndef SGX_STEM_U
#define SGX_STEM_U 4
#endif
    constexpr int U = SGX_STEM_U;                                   // independent elements per thread and round: their loads are in flight together (no divergent control flow)
    const int total = Rin * pitch;
    for (int t0 = tid; t0 < total; t0 += 256 * U) {
        unsigned long long ra[U], rb[U]; SgxDetTab tx[U], ty[U]; int dst[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = min(t0 + 256 * u, total - 1);
            const int ry = (int)sgx_fastdiv((unsigned)t, pitch_magic), col = t - ry * pitch;
            const int iy = iy0 + ry, ix = col - pad;
            const bool inside = (unsigned)iy < (unsigned)T && (unsigned)ix < (unsigned)T;
            ty[u] = ys[ry]; tx[u] = xs[inside ? ix : 0];
            const uint8_t *p0 = base + (size_t)ty[u].o * ipitch + 3 * tx[u].o, *p1 = p0 + ipitch;
            uint32_t a4, b4; uint16_t a2, b2;                      // source pixels tx.o and tx.o + 1 of rows ty.o and ty.o + 1: 6 bytes each (never past the row: tx.o <= W - 2)
            memcpy(&a4, p0, 4); memcpy(&a2, p0 + 4, 2); memcpy(&b4, p1, 4); memcpy(&b2, p1 + 4, 2);
            ra[u] = (unsigned long long)a4 | ((unsigned long long)a2 << 32); rb[u] = (unsigned long long)b4 | ((unsigned long long)b2 << 32);
            dst[u] = t0 + 256 * u < total ? (inside ? ry * pitch + col : -1 - (ry * pitch + col)) : (int)0x80000000;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (dst[u] == (int)0x80000000) continue;
            const bool inside = dst[u] >= 0;
            float *d = tile + (inside ? dst[u] : -1 - dst[u]);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int q00 = (int)((ra[u] >> (8 * c)) & 255u), q01 = (int)((ra[u] >> (8 * c + 24)) & 255u), q10 = (int)((rb[u] >> (8 * c)) & 255u), q11 = (int)((rb[u] >> (8 * c + 24)) & 255u);
                const int h0 = q00 * tx[u].a0 + q01 * tx[u].a1, h1 = q10 * tx[u].a0 + q11 * tx[u].a1;
                const int r = (((ty[u].a0 * (h0 >> 4)) >> 16) + ((ty[u].a1 * (h1 >> 4)) >> 16) + 2) >> 2;
                d[c * plane_stride] = inside ? ((float)(r & 255) - mean[c]) * 1.0f : 0.f;        // outside the T x T input: the convolution's zero padding
            }
        }
    }
    SGX_THREADS_END
    SGX_SYNC();
    const int nbx = (Wo + 3) >> 2;
    SGX_THREADS_BEGIN(tid)
#define SGX_STEM2_RUN(M) sgx_stem2_tasks<3, M>(tid, outc, nrows, nbx, Wo, Ho, r0, pitch, plane_stride, nbx_magic, tile, WtT, bias, out + (size_t)b * out_pitch, (size_t)b * epi.tpitch, epi)
    switch (epi.mode) {
    case SGX_EMODE_NONE: SGX_STEM2_RUN(SGX_EMODE_NONE); break;
    case SGX_EMODE_ACT: SGX_STEM2_RUN(SGX_EMODE_ACT); break;
    default: SGX_STEM2_RUN(SGX_EMODE_HSWISH); break;             // the host launches this kernel for these three programs only
    }
#undef SGX_STEM2_RUN
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_conv_stem: dense K x K convolution with few input channels and <= 16 output channels (the 3 -> 16, 3x3, stride-2 stem).
// A workgroup owns a band of RB output rows of one image: the inc zero-padded input bands are staged in LDS once, every thread
// computes ALL output channels of its pixels (16 accumulators), weights are read with wave-uniform addresses from the
// host-transposed table WtT[(c*K*K + i*K + j)][16].  Accumulation order per output: c, i, j ascending (as k_conv_kxk).
// grid = (nbands, B)
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_conv_stem(int inc, int outc, int H, int W, int Ho, int Wo, int K, int stride, int pad, int RB, unsigned per_magic, unsigned wp_magic, unsigned wo_magic,
                            const float *in, size_t in_pitch, const float *WtT, const float *bias, float *out, size_t out_pitch, SgxEpi epi)
{
    SGX_DYN_LDS(smem);
    float *tile = (float *)smem;
    const int b = (int)blockIdx.y, r0 = (int)blockIdx.x * RB, nrows = min(RB, Ho - r0);
    const int Wp = (Wo - 1) * stride + K, iy0 = r0 * stride - pad, per = ((RB - 1) * stride + K) * Wp;      // full-band plane stride in LDS (also for the last, shorter band)
    SGX_THREADS_BEGIN(tid)
    const int tot = inc * per;
    for (int t0 = tid; t0 < tot; t0 += 256 * 8) {              // 8 independent loads in flight per thread, as k_conv_dw
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = min(t0 + 256 * u, tot - 1);
            const int c = (int)sgx_fastdiv((unsigned)t, per_magic), r = t - c * per;
            const int ry = (int)sgx_fastdiv((unsigned)r, wp_magic), cx = r - ry * Wp;
            const int iy = iy0 + ry, ix = cx - pad;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            v[u] = in[(size_t)b * in_pitch + ((size_t)c * H + (ok ? iy : 0)) * W + (ok ? ix : 0)];
            v[u] = ok ? v[u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int t = t0 + 256 * u; if (t < tot) tile[t] = v[u]; }
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int nout = nrows * Wo;
    for (int o = tid; o < nout; o += 256) {
        const int oy = (int)sgx_fastdiv((unsigned)o, wo_magic), ox = o - oy * Wo;
        float acc[16];
#pragma unroll
        for (int oc = 0; oc < 16; oc++) acc[oc] = oc < outc ? bias[oc] : 0.f;
        for (int c = 0; c < inc; c++) {
            const float *base = tile + c * per + (oy * stride) * Wp + ox * stride;
            for (int i = 0; i < K; i++)
                for (int j = 0; j < K; j++) {
                    const float x = base[i * Wp + j];
                    const float *w = WtT + (size_t)((c * K + i) * K + j) * 16;
#pragma unroll
                    for (int oc = 0; oc < 16; oc++) acc[oc] = fmaf(w[oc], x, acc[oc]);
                }
        }
        const size_t pix = (size_t)(r0 + oy) * Wo + ox;
#pragma unroll
        for (int oc = 0; oc < 16; oc++)
            if (oc < outc) {
                const size_t idx = (size_t)oc * Ho * Wo + pix;
                out[(size_t)b * out_pitch + idx] = sgx_epi(epi, acc[oc], (size_t)b * epi.tpitch + idx);
            }
    }
    SGX_THREADS_END
}

// k_binary: ncnn BinaryOp 0 add / 2 mul / 3 div; b is a same-shape tensor or a scalar (MemoryData w=1)
SGX_KERNEL(256) k_binary(size_t n, int op, const float *a, const float *b, int b_scalar, float bval, float *out)
{
    SGX_THREADS_BEGIN(tid)
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) {
        const float x = a[i], y = b_scalar ? bval : b[i];
        out[i] = op == 0 ? x + y : (op == 2 ? x * y : x / y);
    }
    SGX_THREADS_END
}

// k_unary: Clip / ReLU
SGX_KERNEL(256) k_unary(size_t n, int act, float lo, float hi, const float *a, float *out)
{
    SGX_THREADS_BEGIN(tid)
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) out[i] = sgx_act(a[i], act, lo, hi);
    SGX_THREADS_END
}

// k_permute_hwc_into: ncnn Permute(order 3: c,h,w -> h,w,c) + Flatten + Concat(axis 0): dst[b][off + hw*C + c] = src[b][c][hw]
SGX_KERNEL(256) k_permute_hwc_into(int C, int HW, const float *src, size_t src_pitch, float *dst, size_t dst_pitch, int off)
{
    SGX_THREADS_BEGIN(tid)
    const int idx = (int)blockIdx.x * 256 + tid, b = (int)blockIdx.y;
    if (idx < C * HW) { const int hw = idx / C, c = idx - hw * C; dst[(size_t)b * dst_pitch + off + idx] = src[(size_t)b * src_pitch + (size_t)c * HW + hw]; }
    SGX_THREADS_END
}

// k_copy_into: dst[b][off + i] = src[b][i]
SGX_KERNEL(256) k_copy_into(int n, const float *src, size_t src_pitch, float *dst, size_t dst_pitch, int off)
{
    SGX_THREADS_BEGIN(tid)
    const int idx = (int)blockIdx.x * 256 + tid, b = (int)blockIdx.y;
    if (idx < n) dst[(size_t)b * dst_pitch + off + idx] = src[(size_t)b * src_pitch + idx];
    SGX_THREADS_END
}

// k_softmax_rows: ncnn Softmax over the innermost axis of a (rows x C) blob (mbox_conf_reshape: 2268 x 21): exp(x - max) / sum.
// A workgroup owns 256 consecutive rows: the 256*C floats are staged in LDS with coalesced loads (row stride C is odd -> conflict-free
// per-row walks), every thread normalises its row in LDS, and the block is written back coalesced.  C <= SGX_SOFTMAX_MAXC.
#define SGX_SOFTMAX_MAXC 32
SGX_KERNEL(256) k_softmax_rows(int rows, int C, const float *in, size_t in_pitch, float *out, size_t out_pitch)
{
    SGX_LDS float buf[256 * SGX_SOFTMAX_MAXC + 1];
    const int r0 = (int)blockIdx.x * 256, b = (int)blockIdx.y;
    const int nr = min(256, rows - r0), tot = nr * C;
    const float *src = in + (size_t)b * in_pitch + (size_t)r0 * C;
    float *dst = out + (size_t)b * out_pitch + (size_t)r0 * C;
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < tot; t += 256) buf[t] = src[t];
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid < nr) {
        float *x = buf + tid * C;
        float m = x[0];
        for (int c = 1; c < C; c++) m = fmaxf(m, x[c]);
        float s = 0.f;
        for (int c = 0; c < C; c++) { const float e = expf(x[c] - m); x[c] = e; s += e; }
        for (int c = 0; c < C; c++) x[c] = x[c] / s;
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < tot; t += 256) dst[t] = buf[t];
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_dynamic_mask: the keep/erase predicate of Frame::RmDynamicPointWithSemanticAndGeometry (Frame.cc:556-597):
// epipolar distance of (current keypoint, LK-tracked previous point) under F (CheckEpiLineDistToRmDynamicPoint :613-627, fp64)
// below 0.2 px inside a "person" box (isInDynamicRegion :629-652, strict inequalities) or 1.0 px elsewhere.  keep[i] in {0,1}.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_dynamic_mask(int cap, const uint8_t *keys_raw, const int *n, const float *prev_xy, const double *F, const float *boxes, const int *nboxes,
                               int max_boxes, uint8_t *keep)
{
    SGX_THREADS_BEGIN(tid)
    const int f = (int)blockIdx.y, i = (int)blockIdx.x * 256 + tid;
    if (i < cap) {
        uint8_t k = 0;
        if (i < n[f]) {
            const float *kp = (const float *)(keys_raw + ((size_t)f * cap + i) * 28);
            const float x = kp[0], y = kp[1];
            const double *Fm = F + 9 * (size_t)f;
            const double a = x * Fm[0] + y * Fm[1] + Fm[2], b = x * Fm[3] + y * Fm[4] + Fm[5], c = x * Fm[6] + y * Fm[7] + Fm[8];
            const float px = prev_xy[2 * ((size_t)f * cap + i)], py = prev_xy[2 * ((size_t)f * cap + i) + 1];
            const double dist = fabs(a * px + b * py + c) / sqrt(a * a + b * b);
            // an all-zero F is how sgx_fundamental_ransac_batch_dev reports "cv::findFundamentalMat returned an empty Mat" (fewer than 7 pairs / no
            // model): the reference then reads F12.at<double>() of an empty matrix (undefined behaviour); here such a frame keeps all its keypoints
            const bool noF = Fm[0] == 0. && Fm[1] == 0. && Fm[2] == 0. && Fm[3] == 0. && Fm[4] == 0. && Fm[5] == 0. && Fm[6] == 0. && Fm[7] == 0. && Fm[8] == 0.;
            bool inbox = false;
            const int nb = nboxes[f];
            for (int q = 0; q < nb && q < max_boxes; q++) {
                const float *r = boxes + 4 * ((size_t)f * max_boxes + q);
                if (x > r[0] && x < r[0] + r[2] && y > r[1] && y < r[1] + r[3]) { inbox = true; break; }
            }
            k = (noF || dist < (inbox ? 0.2 : 1.0)) ? 1 : 0;
        }
        keep[(size_t)f * cap + i] = k;
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_compact_keys: the erase step of Frame::RmDynamicPointWithSemanticAndGeometry (Frame.cc:556-604): keypoints with keep == 0 and their
// descriptor rows are removed, order preserved (the reference erases from mvKeys / rebuilds mDescriptors row by row); when a dynamic
// object is present and fewer than 0.1 * nFeatures keypoints survive, everything is restored (:599-604).  One workgroup per frame:
// block scan of the keep flags, then each record (28 B keypoint + 32 B descriptor) moves as dwords.  Out of place.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_compact_keys(int cap, const uint8_t *keys, const uint8_t *desc, const int *n, const uint8_t *keep, const int *have_dynamic, float restore_below,
                               uint8_t *keys_out, uint8_t *desc_out, int *n_out)
{
    SGX_LDS int scan[256];
    SGX_LDS int s_total;
    const int f = (int)blockIdx.x, N = min(n[f], cap), CH = (N + 255) / 256;
    SGX_THREADS_BEGIN(tid)
    int c = 0;
    for (int i = tid * CH; i < min(N, (tid + 1) * CH); i++) c += keep[(size_t)f * cap + i] ? 1 : 0;
    scan[tid] = c;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    sgx_block_exclusive_scan_i32(scan, 256, &s_total, tid);
    SGX_THREADS_END
    SGX_SYNC();
    const bool restore = (have_dynamic && have_dynamic[f]) && (float)s_total < restore_below;       // Cur_keypoint_sum < GetnFeatures()*0.1 (int vs float compare)
    SGX_THREADS_BEGIN(tid)
    int pos = restore ? tid * CH : scan[tid];
    for (int i = tid * CH; i < min(N, (tid + 1) * CH); i++) {
        if (!restore && !keep[(size_t)f * cap + i]) continue;
        const uint32_t *ks = (const uint32_t *)(keys + ((size_t)f * cap + i) * 28), *ds = (const uint32_t *)(desc + ((size_t)f * cap + i) * 32);
        uint32_t *kd = (uint32_t *)(keys_out + ((size_t)f * cap + pos) * 28), *dd = (uint32_t *)(desc_out + ((size_t)f * cap + pos) * 32);
#pragma unroll
        for (int w = 0; w < 7; w++) kd[w] = ks[w];
#pragma unroll
        for (int w = 0; w < 8; w++) dd[w] = ds[w];
        pos++;
    }
    if (tid == 0) n_out[f] = restore ? N : s_total;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// ncnn DetectionOutput (decode + per-class NMS + keep_top_k) and Detector2D::detect's filtering (Detector2D.cc:53-88) on the device.
//   k_det_class_nms  (grid: classes-1 x frames, 256 threads): one (frame, class): candidates = priors with conf > conf_th, sorted by score (descending,
//     ties by prior index — std::stable_sort of the candidate list built in index order), cut to nms_top_k; their boxes are decoded from prior + loc;
//     greedy NMS "keep i unless a kept, higher-ranked box overlaps it by more than nms_th" is resolved exactly: the overlap relation with all higher-ranked
//     candidates is one bit matrix (computed in parallel), the greedy scan then runs 64 candidates at a time (suppression by earlier chunks is a parallel
//     AND with the final kept words; inside a chunk one thread walks 64 rows).  The kept rows are written in rank order.
//   k_det_merge      (grid: frames, 256 threads): the rows of all classes in class-major order, stable-sorted by score, cut to keep_top_k, then the
//     detect() filter (score > det_th, or class 15 "person" above dyn_th; clamp to the 300 x 300 net input, scale to the image) fills the same
//     sgx_det_result the host entry returns and, optionally, the (boxes, count, have-dynamic) arrays sgx_dynamic_mask_batch_dev / compact_keys take.
// Sorting = bitonic network over 64-bit keys (score bits << 32 | ~sequence number) in LDS: score > 0, so its bit pattern orders like the value.
// ---------------------------------------------------------------------------------------------
#define SGX_DO_SORT 4096                   /* per-class sort capacity: num_priors <= this */
#define SGX_DO_TOPK 320                    /* nms_top_k <= this (5 words of 64 candidates) */
#define SGX_DO_WORDS (SGX_DO_TOPK / 64)
#define SGX_DO_MERGE 8192                  /* (classes - 1) * nms_top_k <= this */
struct SgxDetOut { int n, nc, nms_top_k, keep_top_k; float nms_th, conf_th, var0, var1, var2, var3; };

// descending bitonic sort of `size` (power of two) 64-bit keys in LDS by `NT` threads
#define SGX_BITONIC_DESC(keys, size, NT)                                                                             \
    for (int k_ = 2; k_ <= (size); k_ <<= 1)                                                                         \
        for (int j_ = k_ >> 1; j_ > 0; j_ >>= 1) {                                                                   \
            SGX_THREADS_BEGIN(tid)                                                                                   \
            for (int t_ = tid; t_ < (size) / 2; t_ += (NT)) {                                                        \
                const int lo_ = ((t_ & ~(j_ - 1)) << 1) | (t_ & (j_ - 1)), hi_ = lo_ | j_;                           \
                const unsigned long long a_ = keys[lo_], b_ = keys[hi_];                                             \
                const bool desc_ = (lo_ & k_) == 0;                                                                  \
                if (desc_ ? a_ < b_ : a_ > b_) { keys[lo_] = b_; keys[hi_] = a_; }                                   \
            }                                                                                                        \
            SGX_THREADS_END                                                                                          \
            SGX_SYNC();                                                                                              \
        }

// Radix select over the score half of the keys (4 passes of 8 bits, most significant first): finds the score S of the K-th largest key among the non-zero
// keys[0..size) (precondition: at least K of them).  hist/cum: int[256] in LDS; sel: int[4] in LDS = { score bits found so far, keys still needed among the
// current bucket, bucket of this pass, size of the last bucket }.  Afterwards sel[0] = S and "sel[1] == sel[3]" says that exactly K keys have a score >= S
// (no tie across the cut); otherwise the caller falls back to the full sort.  Runs as phases inside a kernel body (256 threads).
#define SGX_TOPK_SELECT(keys, size, K, hist, cum, sel, s_tot)                                                        \
    SGX_THREADS_BEGIN(tid) if (tid == 0) { sel[0] = 0; sel[1] = (K); sel[2] = 0; sel[3] = 0; } SGX_THREADS_END       \
    SGX_SYNC();                                                                                                      \
    for (int sh_ = 24; sh_ >= 0; sh_ -= 8) {                                                                         \
        SGX_THREADS_BEGIN(tid) hist[tid] = 0; SGX_THREADS_END                                                        \
        SGX_SYNC();                                                                                                  \
        SGX_THREADS_BEGIN(tid)                                                                                       \
        const uint32_t pre_ = (uint32_t)sel[0];                                                                      \
        for (int i_ = tid; i_ < (size); i_ += 256) {                                                                 \
            const unsigned long long k_ = keys[i_]; const uint32_t h_ = (uint32_t)(k_ >> 32);                        \
            if (k_ != 0 && (sh_ == 24 || ((h_ ^ pre_) >> ((sh_ + 8) & 31)) == 0)) sgx_atomic_add(&hist[255 - (int)((h_ >> sh_) & 255u)], 1); \
        }                                                                                                            \
        SGX_THREADS_END                                                                                              \
        SGX_SYNC();                                                                                                  \
        SGX_THREADS_BEGIN(tid) cum[tid] = hist[tid]; SGX_THREADS_END                                                 \
        SGX_SYNC();                                                                                                  \
        SGX_THREADS_BEGIN(tid) sgx_block_exclusive_scan_i32(cum, 256, &s_tot, tid); SGX_THREADS_END                  \
        SGX_SYNC();                                                                                                  \
        SGX_THREADS_BEGIN(tid) { const int need_ = sel[1]; if (cum[tid] < need_ && need_ <= cum[tid] + hist[tid]) sel[2] = tid; } SGX_THREADS_END \
        SGX_SYNC();                                                                                                  \
        SGX_THREADS_BEGIN(tid) if (tid == 0) { const int b_ = sel[2]; sel[1] -= cum[b_]; sel[3] = hist[b_]; sel[0] = (int)((uint32_t)sel[0] | ((uint32_t)(255 - b_) << sh_)); } SGX_THREADS_END \
        SGX_SYNC();                                                                                                  \
    }

// rank sort (descending) of the n distinct keys src[0..n) into dst[0..n): rank = number of larger keys.  One pass, no barriers inside; src reads are broadcasts.
#define SGX_RANK_SORT_DESC(src, n, dst)                                                                              \
    SGX_THREADS_BEGIN(tid)                                                                                           \
    for (int r_ = tid; r_ < (n); r_ += 256) {                                                                        \
        const unsigned long long k_ = src[r_]; int rank_ = 0;                                                        \
        _Pragma("unroll 8")                                                                                          \
        for (int q_ = 0; q_ < (n); q_++) rank_ += src[q_] > k_ ? 1 : 0;                                              \
        dst[rank_] = k_;                                                                                             \
    }                                                                                                                \
    SGX_THREADS_END                                                                                                  \
    SGX_SYNC();

// one radix-select pass, resolve step: hist[rb] = keys of the current bucket whose byte is 255 - rb (descending order).  Finds the byte bucket that holds the
// sel[1]-th largest key, updates sel (see SGX_TOPK_SELECT) and clears hist for the next pass.  Called by all threads in one phase; wave 0 works.
#ifdef SGX_EMU
SGX_DEV void sgx_topk_resolve(int *hist, int *sel, int sh, int tid)
{
    if (tid != 0) return;
    int run = 0; const int need = sel[1];
    for (int rb = 0; rb < 256; rb++) {
        const int hcur = hist[rb];
        if (run < need && need <= run + hcur) { sel[1] = need - run; sel[3] = hcur; sel[0] = (int)((uint32_t)sel[0] | ((uint32_t)(255 - rb) << sh)); }
        run += hcur;
    }
    for (int rb = 0; rb < 256; rb++) hist[rb] = 0;
}
#else
SGX_DEV void sgx_topk_resolve(int *hist, int *sel, int sh, int tid)
{
    if (tid >= 64) return;
    const int need = sel[1]; const uint32_t pre = (uint32_t)sel[0];
    int hv[4]; int sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { hv[j] = hist[4 * tid + j]; sum += hv[j]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (tid >= o) inc += t; }
    int run = inc - sum;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (run < need && need <= run + hv[j]) { sel[1] = need - run; sel[3] = hv[j]; sel[0] = (int)(pre | ((uint32_t)(255 - (4 * tid + j)) << sh)); }
        run += hv[j];
        hist[4 * tid + j] = 0;
    }
}
#endif

// Radix select over 32-bit score patterns sc[0..n) (0 = not a candidate; scores are positive floats, so the bit pattern orders like the value): 4 passes of
// 8 bits, most significant first.  Afterwards sel[0] = the K-th largest pattern S, sel[1] = how many of the sel[3] entries equal to S belong to the K largest.
// Precondition: at least K non-zero entries, hist[] all zero (left zero again).  Phases inside a kernel body (256 threads); 2 barriers per pass.
#define SGX_TOPK_SELECT32(sc, n, K, hist, sel)                                                                       \
    SGX_THREADS_BEGIN(tid) if (tid == 0) { sel[0] = 0; sel[1] = (K); sel[2] = 0; sel[3] = 0; } SGX_THREADS_END       \
    SGX_SYNC();                                                                                                      \
    for (int sh_ = 24; sh_ >= 0; sh_ -= 8) {                                                                         \
        SGX_THREADS_BEGIN(tid)                                                                                       \
        const uint32_t pre_ = (uint32_t)sel[0];                                                                      \
        for (int i_ = tid; i_ < (n); i_ += 256) {                                                                    \
            const uint32_t h_ = sc[i_];                                                                              \
            if (h_ != 0 && (sh_ == 24 || ((h_ ^ pre_) >> ((sh_ + 8) & 31)) == 0)) sgx_atomic_add(&hist[255 - (int)((h_ >> sh_) & 255u)], 1); \
        }                                                                                                            \
        SGX_THREADS_END                                                                                              \
        SGX_SYNC();                                                                                                  \
        SGX_THREADS_BEGIN(tid) sgx_topk_resolve(hist, sel, sh_, tid); SGX_THREADS_END                                \
        SGX_SYNC();                                                                                                  \
    }

// "IoU(a, b) > th" exactly as `inter / uni > th` evaluates in fp32, without dividing in the clear cases: th * uni is within 2^-23 (relative) of the real
// product, so an `inter` more than 1e-6 (relative) away from it is on the same side as the correctly rounded quotient; inside that band (and for degenerate
// unions) the division itself decides.
SGX_DEV bool sgx_iou_gt(float a0, float a1, float a2, float a3, float area_a, float b0, float b1, float b2, float b3, float area_b, float th)
{
    const float iw = fminf(a2, b2) - fmaxf(a0, b0), ih = fminf(a3, b3) - fmaxf(a1, b1);
    const float inter = (iw > 0 && ih > 0) ? iw * ih : 0.f;
    const float uni = area_a + area_b - inter;
    if (th > 0.f && uni > 1e-30f && uni < 3e38f) {
        const float pth = th * uni;
        if (inter > pth * 1.000001f) return true;
        if (inter < pth * 0.999999f) return false;
    }
    return inter / uni > th;
}

SGX_KERNEL(256) k_det_class_nms(SgxDetOut P, const float *loc, const float *conf, const float *priors, float *cls_rows, int *cls_count)
{
    SGX_DYN_LDS(sc_pool);                                        // score pattern per prior (u32 view `sc`), 0 = below the confidence threshold: sized by the host to max(num_priors, 7 * SGX_DO_TOPK) words
    unsigned long long *sc64 = (unsigned long long *)sc_pool;     // (a static SGX_DO_SORT-entry array cost 7 KB more than this graph's 2 268 priors need: six workgroups per CU instead of five)
    uint32_t *sc = (uint32_t *)sc_pool;
    // the sorted keys, the decoded boxes and their areas are written only after the last read of `sc` (the selection loop): they live in its storage — 31 KB instead of 40 KB
    // of LDS, five workgroups per CU instead of four (round 4)
    unsigned long long *keys = sc64;                              // [SGX_DO_TOPK]
    float (*box)[4] = (float (*)[4])(sc + 2 * SGX_DO_TOPK);       // [SGX_DO_TOPK][4]
    float *area = (float *)(sc + 6 * SGX_DO_TOPK);                // [SGX_DO_TOPK]
    SGX_LDS unsigned long long over[SGX_DO_TOPK][SGX_DO_WORDS];
    SGX_LDS unsigned long long kept[SGX_DO_WORDS];
    SGX_LDS int hist[256], eqc[256], sel[4];
    SGX_LDS int s_m, s_pos, s_tot;
    unsigned long long *ck = &over[0][0];                        // the unsorted selection lives in the not-yet-used suppression matrix
    const int c = 1 + (int)blockIdx.x, f = (int)blockIdx.y, n = P.n, nc = P.nc;
    const float *L = loc + (size_t)f * n * 4, *C = conf + (size_t)f * n * nc;
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { s_m = 0; s_pos = 0; }
    hist[tid] = 0;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    int cnt = 0;
    for (int i = tid; i < n; i += 256) {
        const float s = C[(size_t)i * nc + c]; uint32_t b = 0;
        if (s > P.conf_th) { memcpy(&b, &s, 4); cnt++; }
        sc[i] = b;
    }
    if (cnt) sgx_atomic_add(&s_m, cnt);
    SGX_THREADS_END
    SGX_SYNC();
    const int ncand = s_m;
    if (ncand == 0) {                                            // a class nobody scored above the threshold (the common case with trained weights): nothing to do
        SGX_THREADS_BEGIN(tid) if (tid == 0) cls_count[f * (nc - 1) + (c - 1)] = 0; SGX_THREADS_END
        return;
    }
    // The NMS only ever looks at the nms_top_k best candidates (ordered by score, ties by prior index — the stable sort of the list built in index order).
    // Select them first (score of rank nms_top_k by radix select; a tie across the cut takes the lowest prior indices), then order just those.
    const int m = min(ncand, P.nms_top_k);
    const int CH = (n + 255) / 256;
    if (ncand > P.nms_top_k) {
        SGX_TOPK_SELECT32(sc, n, P.nms_top_k, hist, sel)
        SGX_THREADS_BEGIN(tid)
        const uint32_t S = (uint32_t)sel[0]; int e = 0;
        for (int i = tid * CH; i < min(n, (tid + 1) * CH); i++) e += sc[i] == S ? 1 : 0;
        eqc[tid] = e;
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid) sgx_block_exclusive_scan_i32(eqc, 256, &s_tot, tid); SGX_THREADS_END
        SGX_SYNC();
    }
    SGX_THREADS_BEGIN(tid)
    const bool cut = ncand > P.nms_top_k;
    const uint32_t S = cut ? (uint32_t)sel[0] : 1u; const int need = cut ? sel[1] : n;
    int run = cut ? eqc[tid] : 0;
    for (int i = tid * CH; i < min(n, (tid + 1) * CH); i++) {
        const uint32_t b = sc[i];
        bool take = b > S || (!cut && b != 0);
        if (cut && b == S) { take = run < need; run++; }
        if (take) ck[sgx_atomic_add(&s_pos, 1)] = ((unsigned long long)b << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_RANK_SORT_DESC(ck, m, keys)
    SGX_THREADS_BEGIN(tid)
    for (int r = tid; r < m; r += 256) {                        // decode (ncnn detectionoutput.cpp; the host code this replaces used the same expressions)
        const int i = (int)(0xFFFFFFFFu - (uint32_t)keys[r]);
        const float *p = priors + 4 * i, *l = L + 4 * i;
        const float pw = p[2] - p[0], ph = p[3] - p[1], pcx = (p[0] + p[2]) * 0.5f, pcy = (p[1] + p[3]) * 0.5f;
        const float cx = P.var0 * l[0] * pw + pcx, cy = P.var1 * l[1] * ph + pcy;
        const float w = expf(P.var2 * l[2]) * pw, hh = expf(P.var3 * l[3]) * ph;
        const float x0 = cx - w * 0.5f, y0 = cy - hh * 0.5f, x1 = cx + w * 0.5f, y1 = cy + hh * 0.5f;
        box[r][0] = x0; box[r][1] = y0; box[r][2] = x1; box[r][3] = y1; area[r] = (x1 - x0) * (y1 - y0);
    }
    for (int w = tid; w < SGX_DO_WORDS; w += 256) kept[w] = 0;
    SGX_THREADS_END
    SGX_SYNC();
    // over[r] bit q: candidate r overlaps the higher-ranked candidate q by more than nms_th (only words up to r / 64 are ever read)
#ifndef SGX_EMU
    {
        const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
        // lanes along the higher-ranked candidates q = 64 w + lane: their boxes do not depend on the row, so every wave keeps them in registers; a row then
        // costs one broadcast read of its own box and one independent compare chain + ballot per word
        float qb[SGX_DO_WORDS][4], qa[SGX_DO_WORDS];
#pragma unroll
        for (int w = 0; w < SGX_DO_WORDS; w++) {
            const int qq = min(64 * w + lane, m - 1);
            qb[w][0] = box[qq][0]; qb[w][1] = box[qq][1]; qb[w][2] = box[qq][2]; qb[w][3] = box[qq][3]; qa[w] = area[qq];
        }
        for (int r = wave; r < m; r += 4) {
            const float a0 = box[r][0], a1 = box[r][1], a2 = box[r][2], a3 = box[r][3], aa = area[r];
#pragma unroll
            for (int w = 0; w < SGX_DO_WORDS; w++) {
                if (64 * w < r) {                                // uniform
                    const bool o = 64 * w + lane < r && sgx_iou_gt(a0, a1, a2, a3, aa, qb[w][0], qb[w][1], qb[w][2], qb[w][3], qa[w], P.nms_th);
                    const unsigned long long bits = __ballot(o);
                    if (lane == 0) over[r][w] = bits;
                }
            }
            if (lane == 0 && (r & 63) == 0) over[r][r >> 6] = 0;    // first row of a chunk: its own word has no higher-ranked member
        }
    }
#else
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < m * SGX_DO_WORDS; t += 256) {
        const int r = t / SGX_DO_WORDS, w = t - r * SGX_DO_WORDS;
        unsigned long long bits = 0;
        for (int b = 0; b < 64; b++) {
            const int q = 64 * w + b;
            if (q >= r) break;
            if (sgx_iou_gt(box[r][0], box[r][1], box[r][2], box[r][3], area[r], box[q][0], box[q][1], box[q][2], box[q][3], area[q], P.nms_th)) bits |= 1ull << b;
        }
        over[r][w] = bits;
    }
    SGX_THREADS_END
#endif
    SGX_SYNC();
    // greedy scan in rank order, 64 candidates (one word) at a time: suppression by earlier words is a parallel AND with their final kept bits, inside the
    // word the decision chain is sequential
#ifndef SGX_EMU
    if ((int)threadIdx.x < 64) {                                 // wave 0: row bits and the "suppressed by an earlier word" flag sit in lanes, the chain runs on the scalar unit
        const int lane = (int)threadIdx.x;
        for (int ch = 0; ch * 64 < m; ch++) {
            const int r = 64 * ch + lane;
            bool s = r >= m;
            if (r < m) for (int w = 0; w < ch; w++) s = s || (over[r][w] & kept[w]) != 0;
            const unsigned long long row = r < m ? over[r][ch] : 0ull;
            const unsigned long long blocked = __ballot(s);
            const uint32_t rlo = (uint32_t)row, rhi = (uint32_t)(row >> 32);
            unsigned long long word = 0;
#pragma unroll
            for (int t = 0; t < 64; t++) {
                const unsigned long long rt = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)rhi, t) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)rlo, t);
                if (!((blocked >> t) & 1) && (rt & word) == 0) word |= 1ull << t;
            }
            if (lane == 0) kept[ch] = word;
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): the next word's lanes read kept[ch]
            __builtin_amdgcn_wave_barrier();
        }
    }
    SGX_SYNC();
#else
    for (int ch = 0; ch * 64 < m; ch++) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) {
            unsigned long long word = 0;
            for (int t = 0; t < 64 && 64 * ch + t < m; t++) {
                const int r = 64 * ch + t; bool s = false;
                for (int w = 0; w < ch; w++) s = s || (over[r][w] & kept[w]) != 0;
                if (!s && (over[r][ch] & word) == 0) word |= 1ull << t;
            }
            kept[ch] = word;
        }
        SGX_THREADS_END
    }
#endif
    SGX_THREADS_BEGIN(tid)
    float *out = cls_rows + ((size_t)f * (nc - 1) + (c - 1)) * SGX_DO_TOPK * 6;
    for (int r = tid; r < m; r += 256) {
        const int w = r >> 6, b = r & 63;
        if (!((kept[w] >> b) & 1)) continue;
        int pos = SGX_POPCLL(kept[w] & ((1ull << b) - 1));
        for (int q = 0; q < w; q++) pos += SGX_POPCLL(kept[q]);
        const uint32_t sb = (uint32_t)(keys[r] >> 32); float s; memcpy(&s, &sb, 4);
        float *o = out + 6 * pos;
        o[0] = (float)c; o[1] = s; o[2] = box[r][0]; o[3] = box[r][1]; o[4] = box[r][2]; o[5] = box[r][3];
    }
    if (tid == 0) { int tot = 0; for (int q = 0; q < SGX_DO_WORDS; q++) tot += SGX_POPCLL(kept[q]); cls_count[f * (nc - 1) + (c - 1)] = tot; }
    SGX_THREADS_END
}

SGX_KERNEL(256) k_det_merge(SgxDetOut P, const float *cls_rows, const int *cls_count, float det_th, float dyn_th, int W, int H, int T,
                            sgx_det_result *results, float *boxes, int *nboxes, int max_boxes, int *have_dynamic)
{
    SGX_LDS unsigned long long keys[SGX_DO_MERGE];
    SGX_LDS unsigned long long top[128];
    SGX_LDS int off[64];
    SGX_LDS int s_total, s_tot, s_pos;
    SGX_LDS int hist[256], cum[256], sel[4];
    const int f = (int)blockIdx.x, ncl = P.nc - 1;
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { int run = 0; for (int q = 0; q < ncl; q++) { off[q] = run; run += cls_count[f * ncl + q]; } off[ncl] = run; s_total = run; }
    SGX_THREADS_END
    SGX_SYNC();
    const int total = s_total;
    int size = 64; while (size < total) size <<= 1;              // <= SGX_DO_MERGE (checked at create)
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < size; i += 256) keys[i] = 0;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int q = 0; q < ncl; q++) {
        const int cnt = off[q + 1] - off[q];
        const float *rows = cls_rows + ((size_t)f * ncl + q) * SGX_DO_TOPK * 6;
        for (int p = tid; p < cnt; p += 256) {
            const float s = rows[6 * p + 1]; uint32_t b; memcpy(&b, &s, 4);
            const int seq = off[q] + p;                          // position in the class-major list the reference sorts (stable: ties keep this order)
            keys[seq] = ((unsigned long long)b << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)seq);
        }
    }
    SGX_THREADS_END
    SGX_SYNC();
    const int K = min(min(total, P.keep_top_k), SGX_DET_MAX);
    // only the K (<= 100) best rows leave the kernel: select them by score, then rank-sort those; a score tie across the cut goes to the full sort
    bool sorted = K == 0;
    if (K > 0 && K <= 128) {
        if (total > K) { SGX_TOPK_SELECT(keys, size, K, hist, cum, sel, s_tot) }
        if (total <= K || sel[1] == sel[3]) {
            SGX_THREADS_BEGIN(tid) if (tid == 0) s_pos = 0; SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            const uint32_t S = total > K ? (uint32_t)sel[0] : 0u;
            for (int i = tid; i < size; i += 256) { const unsigned long long key = keys[i]; if (key != 0 && (uint32_t)(key >> 32) >= S) top[sgx_atomic_add(&s_pos, 1)] = key; }
            SGX_THREADS_END
            SGX_SYNC();
            SGX_RANK_SORT_DESC(top, K, keys)
            sorted = true;
        }
    }
    if (!sorted) { SGX_BITONIC_DESC(keys, size, 256) }
    sgx_det_result *R = results + f;
    SGX_THREADS_BEGIN(tid)
    for (int r = tid; r < K; r += 256) {
        const int seq = (int)(0xFFFFFFFFu - (uint32_t)keys[r]);
        int q = 0; while (q + 1 < ncl && off[q + 1] <= seq) q++;
        const float *row = cls_rows + (((size_t)f * ncl + q) * SGX_DO_TOPK + (seq - off[q])) * 6;
        sgx_detection d; d.label = row[0]; d.score = row[1]; d.xmin = row[2]; d.ymin = row[3]; d.xmax = row[4]; d.ymax = row[5];
        R->raw[r] = d;
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) {                                              // Detector2D::detect, Detector2D.cc:53-88 (sequential: it appends to four lists in row order)
        R->n_raw = K; R->n_objects = 0; R->n_map_boxes = 0; R->n_rm_boxes = 0; R->have_dynamic_for_mapping = 0; R->have_dynamic_for_rm_feature = 0;
        const float Tf = (float)T;
        for (int r = 0; r < K; r++) {
            const sgx_detection v = R->raw[r];
            if (v.score > det_th || (v.score > dyn_th && (int)v.label == 15)) {
                const float x1 = fminf(fmaxf(v.xmin * Tf, 0.f), (float)(T - 1)) / Tf * W, y1 = fminf(fmaxf(v.ymin * Tf, 0.f), (float)(T - 1)) / Tf * H;
                const float x2 = fminf(fmaxf(v.xmax * Tf, 0.f), (float)(T - 1)) / Tf * W, y2 = fminf(fmaxf(v.ymax * Tf, 0.f), (float)(T - 1)) / Tf * H;
                sgx_object2d o; o.id = (int)v.label; o.prob = v.score; o.x = x1; o.y = y1; o.w = x2 - x1; o.h = y2 - y1;
                if (o.id == 15) {
                    R->have_dynamic_for_mapping = 1; if (R->n_map_boxes < SGX_DET_MAX) R->map_boxes[R->n_map_boxes++] = o;
                    if (o.prob > 0.2f) { R->have_dynamic_for_rm_feature = 1; if (R->n_rm_boxes < SGX_DET_MAX) R->rm_boxes[R->n_rm_boxes++] = o; }
                } else if (R->n_objects < SGX_DET_MAX) R->objects[R->n_objects++] = o;
            }
        }
        if (boxes && nboxes) {                                   // the mask stage's inputs: person rectangles (x, y, w, h), their count, the have-dynamic flag
            const int nb = min(R->n_rm_boxes, max_boxes);
            for (int q = 0; q < nb; q++) { float *b = boxes + 4 * ((size_t)f * max_boxes + q); b[0] = R->rm_boxes[q].x; b[1] = R->rm_boxes[q].y; b[2] = R->rm_boxes[q].w; b[3] = R->rm_boxes[q].h; }
            nboxes[f] = nb;
        }
        if (have_dynamic) have_dynamic[f] = R->have_dynamic_for_rm_feature;
    }
    SGX_THREADS_END
}
