// sgx_det_kernels.h — HIP kernels for the 2-D detector forward (MobileNetV3-SSDLite, fp32, NCHW planes like ncnn):
// pre-processing, pointwise convolution as an fp32-MFMA GEMM, depthwise / dense k x k convolution, elementwise ops,
// softmax, layout helpers.  Reference behaviour: src/sg-slam/src/Detector2D.cc:34-45 + the ncnn graph
// src/sg-slam/Thirdparty/ncnn_model/mobilenetv3_ssdlite_voc.param (layer semantics: oracle/detector_oracle.py).
#pragma once
#include "sgx_rt.h"

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
struct SgxEpi { int n; int pad; size_t tpitch; SgxEpiStep s[SGX_EPI_MAX]; };     // tensor operands have the output's shape; per-image pitch tpitch

SGX_DEV float sgx_epi(const SgxEpi &e, float v, size_t toff)
{
    const float root = v;
    for (int i = 0; i < e.n; i++) {
        const int op = e.s[i].op;
        if (op == SGX_EOP_CLIP) v = fminf(fmaxf(v, e.s[i].a), e.s[i].b);
        else if (op == SGX_EOP_RELU) v = fmaxf(v, 0.f);
        else {
            const int src = e.s[i].src;
            const float o = src == SGX_ESRC_CONST ? e.s[i].a : (src == SGX_ESRC_ROOT ? root : e.s[i].t[toff]);
            v = op == SGX_EOP_ADD ? v + o : op == SGX_EOP_MUL ? v * o : op == SGX_EOP_DIV ? v / o : op == SGX_EOP_SUB ? v - o : op == SGX_EOP_RSUB ? o - v : o / v;
        }
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// k_det_preprocess: ncnn::Mat::from_pixels_resize(PIXEL_RGB, w, h, 300, 300) + substract_mean_normalize (Detector2D.cc:39-40).
// ncnn resize_bilinear_c3: 11-bit fixed-point coefficients (host-built tables, clamp to (n-2, 1.0)), then u8 -> f32 - mean.
// out: [B][3][T][T]
// ---------------------------------------------------------------------------------------------
struct SgxDetTab { short o, a0, a1, pad; };

SGX_KERNEL(256) k_det_preprocess(int B, const uint8_t *img, int W, int H, int pitch, const SgxDetTab *xt, const SgxDetTab *yt, int T,
                                 float m0, float m1, float m2, float *out)
{
    SGX_THREADS_BEGIN(tid)
    const int idx = (int)blockIdx.x * 256 + tid, b = (int)blockIdx.y;
    if (idx < T * T) {
        const int y = idx / T, x = idx - y * T;
        const SgxDetTab tx = xt[x], ty = yt[y];
        const uint8_t *r0 = img + ((size_t)b * H + ty.o) * pitch + 3 * tx.o, *r1 = r0 + pitch;
        const float mean[3] = { m0, m1, m2 };
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int h0 = r0[c] * tx.a0 + r0[3 + c] * tx.a1, h1 = r1[c] * tx.a0 + r1[3 + c] * tx.a1;
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
    sgx_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
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
            const float v = sgx_epi(epi, acc[r] + bias[row], (size_t)b * epi.tpitch + (size_t)row * N + col);
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
            float s = 0.f;
            for (int k = 0; k < inc; k++) s = fmaf(Wt[(size_t)row * inc + k], X[(size_t)k * N + col], s);
            const float v = sgx_epi(epi, s + bias[row], (size_t)b * epi.tpitch + (size_t)row * N + col);
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
        float s = 0.f;
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
        out[(size_t)b * out_pitch + (size_t)oc * Ho * Wo + idx] = sgx_epi(epi, s + bias[oc], (size_t)b * epi.tpitch + (size_t)oc * Ho * Wo + idx);
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

// k_softmax_rows: ncnn Softmax over the innermost axis of a (rows x C) blob (mbox_conf_reshape: 2268 x 21): exp(x - max) / sum
SGX_KERNEL(256) k_softmax_rows(int rows, int C, const float *in, size_t in_pitch, float *out, size_t out_pitch)
{
    SGX_THREADS_BEGIN(tid)
    const int r = (int)blockIdx.x * 256 + tid, b = (int)blockIdx.y;
    if (r < rows) {
        const float *x = in + (size_t)b * in_pitch + (size_t)r * C;
        float *y = out + (size_t)b * out_pitch + (size_t)r * C;
        float m = x[0];
        for (int c = 1; c < C; c++) m = fmaxf(m, x[c]);
        float s = 0.f;
        for (int c = 0; c < C; c++) { const float e = expf(x[c] - m); y[c] = e; s += e; }
        for (int c = 0; c < C; c++) y[c] = y[c] / s;
    }
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
            bool inbox = false;
            const int nb = nboxes[f];
            for (int q = 0; q < nb && q < max_boxes; q++) {
                const float *r = boxes + 4 * ((size_t)f * max_boxes + q);
                if (x > r[0] && x < r[0] + r[2] && y > r[1] && y < r[1] + r[3]) { inbox = true; break; }
            }
            k = dist < (inbox ? 0.2 : 1.0) ? 1 : 0;
        }
        keep[(size_t)f * cap + i] = k;
    }
    SGX_THREADS_END
}
