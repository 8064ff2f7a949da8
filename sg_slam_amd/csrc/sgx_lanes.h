// sgx_lanes.h — wave-level kernel vocabulary: per-lane values (vf / vi / vu / vb), the matrix-core operand types and the cross-lane operations a kernel on
// v_mfma_f32_32x32x16_bf16 needs, with two implementations behind ONE kernel source:
//   device (hipcc, gfx950)   vf = float, vi = int, ...; the operations are the hardware instructions (MFMA, v_permlane32_swap, v_pk_fma_f32 with scalar weights, ds_read_b64)
//   emulator (g++ -DSGX_EMU) vf = 64 floats (one per lane of a wave), the kernel body runs once per WAVE, and the cross-lane operations are executed from their lane-layout
//                            definitions: MFMA operands A[i][8 half + j] / B[8 half + j][i], accumulator rows (r & 3) + 8 (r >> 2) + 4 half, permlane32_swap x.hi <-> y.lo.
// The emulator therefore checks the part of such a kernel that the scalar models of rounds 3-5 could not: which lane holds which channel of which pixel, the LDS layout, the
// operand swaps, the host-side weight layout — the things that go wrong on the first run of a matrix-core kernel.  It is test infrastructure (tests/emu/libsgx_emu.so).
// Kernels written against it keep wave-uniform control flow in plain ints and express lane-dependent choices as selects and masked stores.
#pragma once
#include "sgx_rt.h"
#include "sgx_det_block.h"      // sgx_f2, sgx_fma2_w
#include "sgx_det_bf16.h"

#ifndef SGX_EMU
// ------------------------------------------------------------------ device
typedef float vf; typedef int vi; typedef unsigned vu; typedef bool vb;
typedef sgx_f2 vf2;
typedef sgx_f32x16 vf16;
typedef sgx_u32x4 vu4;
typedef sgx_u32x4 sgx_q4;                                                  // 16 bytes of a split-weight operand in memory
struct VB3 { vu4 t0, t1, t2; };
#define SGX_WAVES_BEGIN(w) { const int w = SGX_UNIFORM((int)threadIdx.x >> 6);
#define SGX_WAVES_END }
#define SGX_WAVE_EXIT() return
#define SGX_WPRIV_DECL(type, name, count) type name[count]
#define SGX_WPRIV_BIND(name, w) ((void)0)
SGX_DEV vi v_lane() { return (int)threadIdx.x & 63; }
SGX_DEV vf v_sel(vb c, vf a, vf b) { return c ? a : b; }
SGX_DEV vi v_seli(vb c, vi a, vi b) { return c ? a : b; }
SGX_DEV vi v_min(vi a, vi b) { return a < b ? a : b; }
SGX_DEV vu v_u(vi a) { return (unsigned)a; }
SGX_DEV vf v_fma(vf a, vf b, vf c) { return fmaf(a, b, c); }
SGX_DEV vf v_clip(vf v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
SGX_DEV vf v_clipv(vf v, float lo, vf hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
SGX_DEV vf v_ld(const float *base, vu byte_off) { return *(const float *)((const char *)base + byte_off); }            // uniform base + 32-bit lane offset: global_load with saddr
SGX_DEV void v_st(float *base, vu byte_off, vf v, vb m) { if (m) *(float *)((char *)base + byte_off) = v; }
SGX_DEV vu4 v_ldq(const sgx_q4 *base, vi idx) { return base[idx]; }
SGX_DEV vf2 v_lds_ld2(const sgx_f2 *E, vi idx) { return E[idx]; }
SGX_DEV void v_lds_st2(sgx_f2 *E, vi idx, vf x, vf y, vb m) { if (m) E[idx] = sgx_mk2(x, y); }
SGX_DEV vu4 v_lds_ldq(const sgx_q4 *E, vi idx) { return E[idx]; }
SGX_DEV void v_lds_stq(sgx_q4 *E, vi idx, vu4 v, vb m) { if (m) E[idx] = v; }
SGX_DEV vf v_lds_ld(const float *E, vi idx) { return E[idx]; }
SGX_DEV void v_lds_st(float *E, vi idx, vf v) { E[idx] = v; }
SGX_DEV vf2 v_mk2(vf x, vf y) { return sgx_mk2(x, y); }
SGX_DEV vf v_x(vf2 a) { return a.x; }
SGX_DEV vf v_y(vf2 a) { return a.y; }
SGX_DEV vf2 v_fma2_w(sgx_f2 w, vf2 b, vf2 c) { return sgx_fma2_w(w, b, c); }                                        // c + w * b, w in a scalar register pair
SGX_DEV VB3 v_split3x8(const vf (&v)[8]) { const SgxB3 s = sgx_split3x8(v); VB3 r; r.t0 = s.t0; r.t1 = s.t1; r.t2 = s.t2; return r; }
SGX_DEV vf16 v_mfma3(const vu4 &a0, const vu4 &a1, const vu4 &a2, const VB3 &b, vf16 acc)
{
    acc = SGX_MFMA_BF16(a0, b.t2, acc); acc = SGX_MFMA_BF16(a1, b.t1, acc); acc = SGX_MFMA_BF16(a2, b.t0, acc);
    acc = SGX_MFMA_BF16(a0, b.t1, acc); acc = SGX_MFMA_BF16(a1, b.t0, acc);
    return SGX_MFMA_BF16(a0, b.t0, acc);
}
// (x, y) -> (x with its upper half-wave replaced by y's lower, y with its lower half-wave replaced by x's upper): v_permlane32_swap
SGX_DEV void v_swap32(vf x, vf y, vf &nx, vf &ny)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    nx = __uint_as_float(r[0]); ny = __uint_as_float(r[1]);
}
#else
// ------------------------------------------------------------------ emulator: a value per lane of one wave
template <class T> struct sgx_lv {
    T v[64];
    sgx_lv() { for (int l = 0; l < 64; l++) v[l] = T(); }
    sgx_lv(T s) { for (int l = 0; l < 64; l++) v[l] = s; }
};
typedef sgx_lv<float> vf; typedef sgx_lv<int> vi; typedef sgx_lv<unsigned> vu; typedef sgx_lv<bool> vb;
#define SGX_LV_BIN(op, R)                                                                                                                                   \
    template <class T> static inline sgx_lv<R> operator op(const sgx_lv<T> &a, const sgx_lv<T> &b) { sgx_lv<R> r; for (int l = 0; l < 64; l++) r.v[l] = (R)(a.v[l] op b.v[l]); return r; } \
    template <class T> static inline sgx_lv<R> operator op(const sgx_lv<T> &a, T b) { sgx_lv<R> r; for (int l = 0; l < 64; l++) r.v[l] = (R)(a.v[l] op b); return r; }                  \
    template <class T> static inline sgx_lv<R> operator op(T a, const sgx_lv<T> &b) { sgx_lv<R> r; for (int l = 0; l < 64; l++) r.v[l] = (R)(a op b.v[l]); return r; }
#define SGX_LV_ARITH(op) SGX_LV_BIN(op, T)
SGX_LV_ARITH(+) SGX_LV_ARITH(-) SGX_LV_ARITH(*) SGX_LV_ARITH(/) SGX_LV_ARITH(&) SGX_LV_ARITH(|) SGX_LV_ARITH(>>) SGX_LV_ARITH(<<)
SGX_LV_BIN(<, bool) SGX_LV_BIN(<=, bool) SGX_LV_BIN(>, bool) SGX_LV_BIN(>=, bool) SGX_LV_BIN(==, bool) SGX_LV_BIN(!=, bool)
static inline vb operator!(const vb &a) { vb r; for (int l = 0; l < 64; l++) r.v[l] = !a.v[l]; return r; }
struct vf2 { vf x, y; };
struct vf16 { vf r[16]; vf &operator[](int i) { return r[i]; } const vf &operator[](int i) const { return r[i]; } };
struct vu4 { vu c[4]; vu &operator[](int i) { return c[i]; } const vu &operator[](int i) const { return c[i]; } };
struct sgx_q4 { unsigned v[4]; };
struct VB3 { vu4 t0, t1, t2; };
#define SGX_WAVES_BEGIN(w) for (int w = 0; w < (int)blockDim.x / 64; ++w) {
#define SGX_WAVES_END }
#define SGX_WAVE_EXIT() continue
#define SGX_WPRIV_DECL(type, name, count) static thread_local type name##_store[16][count]
#define SGX_WPRIV_BIND(name, w) auto *name = name##_store[w]
static inline vi v_lane() { vi r; for (int l = 0; l < 64; l++) r.v[l] = l; return r; }
static inline vf v_sel(const vb &c, const vf &a, const vf &b) { vf r; for (int l = 0; l < 64; l++) r.v[l] = c.v[l] ? a.v[l] : b.v[l]; return r; }
static inline vi v_seli(const vb &c, const vi &a, const vi &b) { vi r; for (int l = 0; l < 64; l++) r.v[l] = c.v[l] ? a.v[l] : b.v[l]; return r; }
static inline vi v_min(const vi &a, const vi &b) { vi r; for (int l = 0; l < 64; l++) r.v[l] = a.v[l] < b.v[l] ? a.v[l] : b.v[l]; return r; }
static inline vu v_u(const vi &a) { vu r; for (int l = 0; l < 64; l++) r.v[l] = (unsigned)a.v[l]; return r; }
static inline vf v_fma(const vf &a, const vf &b, const vf &c) { vf r; for (int l = 0; l < 64; l++) r.v[l] = fmaf(a.v[l], b.v[l], c.v[l]); return r; }
static inline vf v_clip(const vf &v, float lo, float hi) { vf r; for (int l = 0; l < 64; l++) r.v[l] = fminf(fmaxf(v.v[l], lo), hi); return r; }
static inline vf v_clipv(const vf &v, float lo, const vf &hi) { vf r; for (int l = 0; l < 64; l++) r.v[l] = fminf(fmaxf(v.v[l], lo), hi.v[l]); return r; }
static inline vf v_ld(const float *base, const vu &off) { vf r; for (int l = 0; l < 64; l++) r.v[l] = *(const float *)((const char *)base + off.v[l]); return r; }
static inline void v_st(float *base, const vu &off, const vf &v, const vb &m) { for (int l = 0; l < 64; l++) if (m.v[l]) *(float *)((char *)base + off.v[l]) = v.v[l]; }
static inline vu4 v_ldq(const sgx_q4 *base, const vi &idx) { vu4 r; for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) r.c[j].v[l] = base[idx.v[l]].v[j]; return r; }
static inline vf2 v_lds_ld2(const sgx_f2 *E, const vi &idx) { vf2 r; for (int l = 0; l < 64; l++) { r.x.v[l] = E[idx.v[l]].x; r.y.v[l] = E[idx.v[l]].y; } return r; }
static inline void v_lds_st2(sgx_f2 *E, const vi &idx, const vf &x, const vf &y, const vb &m) { for (int l = 0; l < 64; l++) if (m.v[l]) { E[idx.v[l]].x = x.v[l]; E[idx.v[l]].y = y.v[l]; } }
static inline vu4 v_lds_ldq(const sgx_q4 *E, const vi &idx) { return v_ldq(E, idx); }
static inline void v_lds_stq(sgx_q4 *E, const vi &idx, const vu4 &v, const vb &m) { for (int l = 0; l < 64; l++) if (m.v[l]) for (int j = 0; j < 4; j++) E[idx.v[l]].v[j] = v.c[j].v[l]; }
static inline vf v_lds_ld(const float *E, const vi &idx) { vf r; for (int l = 0; l < 64; l++) r.v[l] = E[idx.v[l]]; return r; }
static inline void v_lds_st(float *E, const vi &idx, const vf &v) { for (int l = 0; l < 64; l++) E[idx.v[l]] = v.v[l]; }
static inline vf2 v_mk2(const vf &x, const vf &y) { vf2 r; r.x = x; r.y = y; return r; }
static inline vf v_x(const vf2 &a) { return a.x; }
static inline vf v_y(const vf2 &a) { return a.y; }
static inline vf2 v_fma2_w(sgx_f2 w, const vf2 &b, const vf2 &c) { vf2 r; for (int l = 0; l < 64; l++) { r.x.v[l] = fmaf(w.x, b.x.v[l], c.x.v[l]); r.y.v[l] = fmaf(w.y, b.y.v[l], c.y.v[l]); } return r; }
// the exact three-term bf16 split of sgx_split3 (round to nearest even, exact residuals), element j of a lane's eight in bits 16 (j & 1) of register j >> 1
static inline VB3 v_split3x8(const vf (&v)[8])
{
    VB3 b;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) {
        const float x = v[j].v[l];
        const unsigned short h0 = sgx_bf16_rne(x); const float r1 = x - sgx_bf16_to_f32(h0);
        const unsigned short h1 = sgx_bf16_rne(r1); const unsigned short h2 = sgx_bf16_rne(r1 - sgx_bf16_to_f32(h1));
        const int sh = 16 * (j & 1); const unsigned keep = ~(0xffffu << sh);
        b.t0.c[j >> 1].v[l] = (b.t0.c[j >> 1].v[l] & keep) | ((unsigned)h0 << sh);
        b.t1.c[j >> 1].v[l] = (b.t1.c[j >> 1].v[l] & keep) | ((unsigned)h1 << sh);
        b.t2.c[j >> 1].v[l] = (b.t2.c[j >> 1].v[l] & keep) | ((unsigned)h2 << sh);
    }
    return b;
}
// one v_mfma_f32_32x32x16_bf16 from its lane layout: lane (half, i) of A holds A[i][8 half + j], of B holds B[8 half + j][i], j = 0..7 (bf16 j in bits 16 (j & 1) of register
// j >> 1); lane (half, i) of C / D holds column i, rows (r & 3) + 8 (r >> 2) + 4 half.  The sixteen exact products of an output are summed wide and rounded into the fp32
// accumulator once (the hardware's internal order is unspecified; any fp32-grade order meets the parity criteria).
static inline float sgx_lv_bf16(const vu4 &a, int lane, int j) { return sgx_bf16_to_f32((unsigned short)(a.c[j >> 1].v[lane] >> (16 * (j & 1)))); }
static inline vf16 v_mfma1(const vu4 &a, const vu4 &b, const vf16 &c)
{
    vf16 d;
    for (int half = 0; half < 2; half++) for (int i = 0; i < 32; i++) for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        double sum = 0;
        for (int k = 0; k < 16; k++) sum += (double)sgx_lv_bf16(a, 32 * (k >> 3) + row, k & 7) * (double)sgx_lv_bf16(b, 32 * (k >> 3) + i, k & 7);
        d.r[r].v[32 * half + i] = (float)((double)c.r[r].v[32 * half + i] + sum);
    }
    return d;
}
static inline vf16 v_mfma3(const vu4 &a0, const vu4 &a1, const vu4 &a2, const VB3 &b, vf16 acc)
{
    acc = v_mfma1(a0, b.t2, acc); acc = v_mfma1(a1, b.t1, acc); acc = v_mfma1(a2, b.t0, acc);
    acc = v_mfma1(a0, b.t1, acc); acc = v_mfma1(a1, b.t0, acc);
    return v_mfma1(a0, b.t0, acc);
}
static inline void v_swap32(const vf &x, const vf &y, vf &nx, vf &ny)
{
    vf a, b;
    for (int l = 0; l < 32; l++) { a.v[l] = x.v[l]; a.v[32 + l] = y.v[l]; b.v[l] = x.v[32 + l]; b.v[32 + l] = y.v[32 + l]; }
    nx = a; ny = b;
}
#endif
