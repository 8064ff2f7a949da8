// sgx_det.cpp — host side of the 2-D detector C-ABI: ncnn .param/.bin loader, shape inference, execution plan
// (with conv+activation and Permute/Flatten/Concat fusion), batched forward, DetectionOutput + Detector2D::detect
// post-processing, dynamic-feature mask.  Reference: src/sg-slam/src/Detector2D.cc:16-89, src/sg-slam/src/Frame.cc:556-604.
#include "sgx_block.h"
#include "sgx_det_block.h"
#include "sgx_det_irb.h"
#include "sgx_det_bf16.h"
#include "sgx_det_hrb.h"
#include "sgx_prof.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

namespace {
struct Layer {
    std::string type, name;
    std::vector<std::string> ins, outs;
    std::map<int, double> p;
    std::map<int, std::vector<float>> pa;
    int geti(int k, int d) const { auto it = p.find(k); return it == p.end() ? d : (int)it->second; }
    float getf(int k, float d) const { auto it = p.find(k); return it == p.end() ? d : (float)it->second; }
};
struct Blob { int c = 0, h = 0, w = 0; size_t n = 0; float *d = nullptr; bool scalar = false; float sval = 0.f; int alias = -1; };
enum OpKind { OP_PW, OP_KXK, OP_BINARY, OP_UNARY, OP_PERMUTE_INTO, OP_COPY_INTO, OP_SOFTMAX, OP_FUSED_BLOCK, OP_IRB, OP_SE_GATE };
struct EpiStep { int op, src; float a, b; int tensor; };
struct Op {
    OpKind kind; int in0 = -1, in1 = -1, out = -1;
    std::vector<EpiStep> epi; int hwc = 0, hwc_off = 0; bool dead = false; std::string name;
    int inc = 0, outc = 0, H = 0, W = 0, Ho = 0, Wo = 0, k = 1, stride = 1, pad = 0, depthwise = 0, act = 0; float lo = 0, hi = 0;
    float *wt = nullptr, *bias = nullptr, *wtT = nullptr, *wtP = nullptr; int ldw = 0;      // wtP: depthwise weights with the channels of a pair interleaved [C / 2][k * k][2] (k_conv_dw3)
    int bop = 0; int off = 0; int rows = 0, C = 0;
    void *wS = nullptr;                         // pointwise weights split into three bf16 terms in the MFMA operand layout (sgx_det_bf16.h), ld = ldw; NULL in the exact-fp32 plan
    SgxFusedBlk fb; int fb_res_blob = -1;      // OP_FUSED_BLOCK: expand -> depthwise -> project (+ residual) in one kernel (sgx_det_block.h)
    SgxSeGate sg; int sg_res_blob = -1;                            // OP_SE_GATE: squeeze -> excite -> gate x input [+ residual] as one kernel (k_se_gate, sgx_det_block.h)
    SgxIrb irb; int irb_res_blob = -1, irb_out2_blob = -1;         // OP_IRB: [expand ->] depthwise -> project [-> squeeze-excite gate] [+ residual] on the matrix cores (sgx_det_irb.h)
};
}  // namespace

// Capture lanes of the plan's hipGraph (capture_forked) and executable instances used in turn.  DEFAULT 1 / 1 = the one-stream chain of rounds 1-4.  Round 5 built the forked
// capture and measured it (profiles/r5_ab_concurrency.md): with 3 lanes the detector stream's span per step falls 17.4 -> 16.0 ms and the tracking stream's matchers run at their
// stand-alone times, throughput unchanged — but on this runtime (ROCm 7.2) launching a graph WITH parallel branches blocks the calling host thread until the graph has run, so the
// host-input path loses its copy / compute overlap (26.6 k -> 15.9 k frames/s, 40.8 -> 24.4 GB/s of PCIe; two instances used in turn recover only part: 18.6 k).  A gain that is a
// side effect of a host-side serialisation is not kept as the default; SGX_DET_FORK / SGX_DET_EXECS remain as taps of the tap build.
#ifndef SGX_DET_GRAPH_EXECS
#define SGX_DET_GRAPH_EXECS 1
#endif
#ifndef SGX_DET_FORK_LANES
#define SGX_DET_FORK_LANES 1
#endif

struct sgx_det {
    int T = 300, max_batch = 1, W = 0, H = 0, legacy = 0;
    int gemm = 0;                       // matrix products of the pointwise convolutions: 0 exact fp32 (v_mfma_f32_32x32x2_f32: an ascending-k fmaf chain), 1 bf16x3 (sgx_det_bf16.h)
    int pre_fused = 0;                  // 1: ops[0] is the stem convolution and runs as k_stem_pre straight from the u8 images (no "input" blob, no k_det_preprocess launch)
    float det_th = 0.9f, dyn_th = 0.01f;
    std::vector<Layer> layers;
    std::map<std::string, int> blob_id;
    std::vector<Blob> blobs;
    std::vector<Op> ops;
    std::vector<void *> dev;
    std::vector<float> priors;          // 2 x (num_priors*4): boxes, variances (PriorBox is input-independent: host constant)
    int num_priors = 0, num_class = 21, nms_top_k = 300, keep_top_k = 100; float nms_th = 0.45f, conf_th = 0.01f, dvar[4] = {0.1f, 0.1f, 0.2f, 0.2f};
    int loc_blob = -1, conf_blob = -1;
    SgxDetTab *d_xt = nullptr, *d_yt = nullptr; uint8_t *d_img = nullptr;
    float *d_priors = nullptr, *d_cls_rows = nullptr; int *d_cls_count = nullptr; sgx_det_result *d_results = nullptr;      // DetectionOutput on the device
    double gmac = 0;
#ifndef SGX_EMU
    std::map<int, std::vector<hipGraphExec_t>> graphs;     // captured plan per batch size (launch-bound tail of ~100 small kernels -> one graph launch); SGX_DET_EXECS instances used in turn
    std::map<int, unsigned> graph_turn;
    std::vector<hipStream_t> fork_streams;    // side lanes of the forked capture (capture_forked): they only exist to give the graph its parallel branches
    std::vector<hipEvent_t> fork_events;
#endif
    ~sgx_det() {
#ifndef SGX_EMU
        for (auto &g : graphs) for (hipGraphExec_t e : g.second) (void)hipGraphExecDestroy(e);
        for (hipEvent_t e : fork_events) (void)hipEventDestroy(e);
        for (hipStream_t q : fork_streams) (void)hipStreamDestroy(q);
#endif
        for (void *p : dev) (void)hipFree(p);
    }
    template <class Tp> int alloc(Tp **p, size_t n) { void *q = nullptr; if (hipMalloc(&q, (n ? n : 1) * sizeof(Tp)) != hipSuccess) return SGX_ERR_NOMEM; dev.push_back(q); *p = (Tp *)q; return SGX_OK; }
};

static thread_local int g_det_fuse = 1;
static thread_local int g_det_block_fusion = 0;       // test / tuning tap (read at sgx_det_create): fuse expand -> depthwise -> project triples into k_fused_block
static thread_local int g_det_legacy = 0;             // test tap (read at sgx_det_create): run the simple reference kernels (k_conv_pw / k_conv_kxk) instead of the tuned ones
SGX_TAP int sgx_det_debug_set_fusion(int on) { g_det_fuse = on ? 1 : 0; return SGX_OK; }
SGX_TAP int sgx_det_debug_set_legacy_kernels(int on) { g_det_legacy = on ? 1 : 0; return SGX_OK; }
SGX_TAP int sgx_det_debug_set_block_fusion(int on) { g_det_block_fusion = on ? 1 : 0; return SGX_OK; }
static thread_local int g_det_irb = -1;               // test / tuning tap: inverted-residual blocks and SSD heads as one matrix-core kernel each (sgx_det_irb.h): 0 off, 1 the shapes where it beats
                                         // the per-layer kernels on MI355X (default), 2 every shape it supports (tests); -1 = SGX_DET_IRB or the default
SGX_TAP int sgx_det_debug_set_irb(int on) { g_det_irb = on < 0 ? -1 : (on > 2 ? 2 : on); return SGX_OK; }
// Matrix-product scheme of the pointwise / expand / project / squeeze-excite convolutions (read at sgx_det_create): 0 = exact fp32 on v_mfma_f32_32x32x2_f32 (bit-identical to the
// per-layer reference kernels and to the emulator: the anchor of the plan-equality tests), 1 = bf16x3 on v_mfma_f32_32x32x16_bf16 (fp32-accurate products, fp32 accumulation,
// another summation order; sgx_det_bf16.h), -1 = SGX_DET_GEMM (f32 | bf16x3) or the default.  The emulator build always runs 0.
static thread_local int g_det_gemm = -1;
#ifndef SGX_DET_GEMM_DEFAULT
#define SGX_DET_GEMM_DEFAULT 1
#endif
SGX_TAP int sgx_det_debug_set_gemm(int mode) { g_det_gemm = mode < 0 ? -1 : (mode ? 1 : 0); return SGX_OK; }
extern "C" int sgx_det_gemm_mode(const sgx_det *h) { return h ? h->gemm : -1; }
static int det_gemm_mode()
{
    // the emulator build defaults to the exact-fp32 plan (its kernels ARE the ascending-k fmaf chains); asked for bf16x3 it runs the pointwise layers through a software model of
    // k_conv_pw3 (sgx_pw3_emu below) so that the planner's bf16x3 branch — split weights, their layout and padding, the k >= 64 rule — is covered by the CPU tier
    if (g_det_gemm >= 0) return g_det_gemm;
    const char *e = sgx_getenv("SGX_DET_GEMM");
    if (e && *e) return (!strcmp(e, "f32") || !strcmp(e, "0")) ? 0 : 1;
#ifdef SGX_EMU
    return 0;
#else
    return SGX_DET_GEMM_DEFAULT;
#endif
}

// the 3 x 3 stride-2 stem on the k_conv_stem2 path (run_op) — also the condition for fusing the pre-processing into it (k_stem_pre)
struct Stem2Geom { int nbx4, pitch4, RB, nbands; size_t lds; };
static Stem2Geom stem2_geom(const Op &op)
{
    Stem2Geom g;
    g.nbx4 = (op.Wo + 3) / 4; g.pitch4 = ((g.nbx4 - 1) * 4 * op.stride + 3 * op.stride + op.k + 3) & ~3;
    static const int env_words = sgx_getenv("SGX_DET_STEM_WORDS") ? atoi(sgx_getenv("SGX_DET_STEM_WORDS")) : 13824;      // tuning tap: LDS budget of the band tile in floats.  Round 6 sweep at 512 frames (k_stem_pre, ms): 4608 0.66, 6400 0.57, 8192 0.47, 10240 0.48-0.49 (rounds 3-6), 12000 0.49, 13824 0.44-0.45 (seven output rows per band), 17408 0.60, 24000 0.83
    g.RB = std::max(1, std::min(op.Ho, ((env_words / (3 * g.pitch4)) - 3) / 2 + 1));       // LDS = 3 x ((RB - 1) 2 + 3) x pitch4 floats, about 54 KB
    g.nbands = (op.Ho + g.RB - 1) / g.RB;
    g.lds = (size_t)3 * ((g.RB - 1) * 2 + 3) * g.pitch4 * 4;
    return g;
}
static bool stem2_epi_ok(const sgx_det *h, const Op &op);
static bool stem2_ok(const sgx_det *h, const Op &op)
{
    static const int dw2_on = sgx_getenv("SGX_DW2") ? atoi(sgx_getenv("SGX_DW2")) : 1;
    if (op.kind != OP_KXK || !dw2_on || h->legacy || op.depthwise || !op.wtT || op.outc > 16 || op.inc != 3 || op.k != 3 || op.stride != 2) return false;
    const Stem2Geom g = stem2_geom(op);
    return 3 * 5 * g.pitch4 * 4 <= 65536 && g.pitch4 < 4096 && stem2_epi_ok(h, op);
}

static int parse_param(const char *text, std::vector<Layer> &layers)
{
    std::istringstream is(text);
    std::string line;
    if (!std::getline(is, line) || atoi(line.c_str()) != 7767517) return SGX_ERR_INVALID;
    if (!std::getline(is, line)) return SGX_ERR_INVALID;
    while (std::getline(is, line)) {
        std::istringstream ls(line);
        Layer L; int nin = 0, nout = 0;
        if (!(ls >> L.type >> L.name >> nin >> nout)) continue;
        for (int i = 0; i < nin; i++) { std::string s; ls >> s; L.ins.push_back(s); }
        for (int i = 0; i < nout; i++) { std::string s; ls >> s; L.outs.push_back(s); }
        std::string kv;
        while (ls >> kv) {
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) continue;
            const int k = atoi(kv.substr(0, eq).c_str());
            const std::string v = kv.substr(eq + 1);
            if (k <= -23300) {
                std::vector<float> arr; std::istringstream vs(v); std::string tok; bool first = true;
                while (std::getline(vs, tok, ',')) { if (first) { first = false; continue; } arr.push_back((float)atof(tok.c_str())); }
                L.pa[-k - 23300] = arr;
            } else L.p[k] = atof(v.c_str());
        }
        layers.push_back(L);
    }
    return layers.empty() ? SGX_ERR_INVALID : SGX_OK;
}

static void build_tab(int s, int d, std::vector<SgxDetTab> &t)
{   // ncnn resize_bilinear (mat_pixel_resize.cpp): clamp to (s-2, 1.0) at the far edge
    t.resize(d);
    const double scale = (double)s / d;
    for (int i = 0; i < d; i++) {
        float f = (float)((i + 0.5) * scale - 0.5);
        int si = (int)floorf(f); f -= si;
        if (si < 0) { si = 0; f = 0.f; }
        if (si >= s - 1) { si = s - 2; f = 1.f; }
        t[i].o = (short)si; t[i].a0 = (short)lrintf((1.f - f) * 2048); t[i].a1 = (short)lrintf(f * 2048); t[i].pad = 0;
    }
}

extern "C" void sgx_det_destroy(sgx_det *h) { delete h; }

extern "C" int sgx_det_create(const char *param_text, const void *bin, size_t bin_bytes, int width, int height, int max_batch,
                              float detection_confidence_threshold, float dynamic_detection_confidence_threshold, sgx_det **out)
{
    if (!param_text || !bin || !out || width < 8 || height < 8 || width > SGX_PRE_MAXW || max_batch < 1) return SGX_ERR_INVALID;
    sgx_det *h = new sgx_det();
    h->W = width; h->H = height; h->max_batch = max_batch; h->legacy = g_det_legacy; h->gemm = g_det_legacy ? 0 : det_gemm_mode(); h->det_th = detection_confidence_threshold; h->dyn_th = dynamic_detection_confidence_threshold;
    int rc = parse_param(param_text, h->layers);
    if (rc != SGX_OK) { delete h; return rc; }
    const int B = max_batch, T = h->T;
    const uint8_t *bp = (const uint8_t *)bin; size_t bo = 0;
    auto blob = [&](const std::string &n) -> int { auto it = h->blob_id.find(n); if (it != h->blob_id.end()) return it->second; h->blob_id[n] = (int)h->blobs.size(); h->blobs.push_back(Blob()); return (int)h->blobs.size() - 1; };
    auto resolve = [&](int id) { while (h->blobs[id].alias >= 0) id = h->blobs[id].alias; return id; };
#define FAIL(code) do { delete h; return (code); } while (0)
    std::map<std::string, std::pair<int, int>> into;       // conv output name -> (concat blob id, offset): fused Permute+Flatten+Concat
    std::map<std::string, int> concat_off;                 // flatten output name -> offset inside its concat
    std::vector<float> prior_boxes, prior_vars;
    for (size_t li = 0; li < h->layers.size(); li++) {
        const Layer &L = h->layers[li];
        if (L.type == "Input") { Blob &b = h->blobs[blob(L.outs[0])]; b.c = 3; b.h = T; b.w = T; b.n = (size_t)3 * T * T; if (h->alloc(&b.d, b.n * B)) FAIL(SGX_ERR_NOMEM); continue; }
        if (L.type == "MemoryData") {
            const int n = L.geti(0, 0) * std::max(L.geti(1, 1), 1) * std::max(L.geti(2, 1), 1);
            if (n != 1 || bo + 4 > bin_bytes) FAIL(SGX_ERR_UNSUPPORTED);
            Blob &b = h->blobs[blob(L.outs[0])]; b.scalar = true; memcpy(&b.sval, bp + bo, 4); bo += 4; b.n = 1; continue;
        }
        if (L.type == "Split") { const int src = blob(L.ins[0]); for (const std::string &o : L.outs) { const int id = blob(o); h->blobs[id] = h->blobs[resolve(src)]; h->blobs[id].alias = resolve(src); } continue; }
        const int in0 = resolve(blob(L.ins[0]));
        const Blob A = h->blobs[in0];
        if (L.type == "Convolution" || L.type == "ConvolutionDepthWise") {
            Op op; const int outc = L.geti(0, 0), k = L.geti(1, 1), stride = L.geti(3, 1), pad = L.geti(4, 0), wsize = L.geti(6, 0), group = L.geti(7, 1);
            const int inc = A.c;
            if (L.geti(2, 1) != 1 || (group != 1 && group != inc) || wsize != outc * (inc / group) * k * k) FAIL(SGX_ERR_UNSUPPORTED);
            if (bo + 4 + (size_t)wsize * 4 + (L.geti(5, 0) ? (size_t)outc * 4 : 0) > bin_bytes) FAIL(SGX_ERR_INVALID);
            uint32_t flag; memcpy(&flag, bp + bo, 4); bo += 4;
            if (flag != 0) FAIL(SGX_ERR_UNSUPPORTED);                       // raw fp32 weights only
            if (h->alloc(&op.wt, wsize) || h->alloc(&op.bias, outc)) FAIL(SGX_ERR_NOMEM);
            if (hipMemcpy(op.wt, bp + bo, (size_t)wsize * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
            {   // host-transposed copies for the tuned kernels: pointwise [k][oc]; stem [(c,i,j)][16]
                const float *wsrc = (const float *)(bp + bo);
                const bool pw = (k == 1 && group == 1 && stride == 1 && pad == 0), stem = (group == 1 && k > 1 && outc <= 16);
                if (pw || stem) {
                    const int kk = inc * k * k, ldo = pw ? ((outc + 31) / 32) * 32 + 128 : 16;          // pointwise: zero-padded to [ceil32(inc)][ldo] (ldo covers the last oc block of every tile shape: up to four padding tiles of 32)
                    op.ldw = ldo;
                    std::vector<float> wT((size_t)(pw ? ((kk + 31) / 32) * 32 : kk) * ldo, 0.f);
                    for (int o = 0; o < outc; o++) for (int q = 0; q < kk; q++) { float v; memcpy(&v, wsrc + (size_t)o * kk + q, 4); wT[(size_t)q * ldo + o] = v; }
                    if (h->alloc(&op.wtT, wT.size())) FAIL(SGX_ERR_NOMEM);
                    if (hipMemcpy(op.wtT, wT.data(), wT.size() * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                    if (pw && h->gemm == 1) {                           // bf16x3 plan: the same weights as three bf16 terms in the matrix-core operand layout
                        std::vector<float> wf((size_t)outc * kk); memcpy(wf.data(), wsrc, wf.size() * 4);
                        std::vector<unsigned short> ws((size_t)((kk + 15) / 16) * 6 * ldo * 8);
                        sgx_split_weights_bf16x3(wf.data(), outc, kk, ldo, ws.data());
                        unsigned short *dws = nullptr;
                        if (h->alloc(&dws, ws.size())) FAIL(SGX_ERR_NOMEM);
                        if (hipMemcpy(dws, ws.data(), ws.size() * 2, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                        op.wS = dws;
                    }
                }
            }
#ifdef SGX_DEBUG_TAPS      // k_conv_dw3 (round-6 experiment, slower than k_conv_dw2) exists in the tap build only
            if (group == inc && group != 1 && (outc & 1) == 0 && inc == outc) {      // depthwise: pair-interleaved copy for k_conv_dw3
                const float *wsrc = (const float *)(bp + bo); const int kk = k * k;
                std::vector<float> w2((size_t)outc * kk);
                for (int m = 0; m < outc; m++) for (int t = 0; t < kk; t++) memcpy(&w2[((size_t)(m >> 1) * kk + t) * 2 + (m & 1)], wsrc + (size_t)m * kk + t, 4);
                if (h->alloc(&op.wtP, w2.size())) FAIL(SGX_ERR_NOMEM);
                if (hipMemcpy(op.wtP, w2.data(), w2.size() * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
            }
#endif
            bo += (size_t)wsize * 4;
            std::vector<float> bz(outc, 0.f);
            if (L.geti(5, 0)) { memcpy(bz.data(), bp + bo, (size_t)outc * 4); bo += (size_t)outc * 4; }
            if (hipMemcpy(op.bias, bz.data(), (size_t)outc * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
            op.inc = inc; op.outc = outc; op.H = A.h; op.W = A.w; op.k = k; op.stride = stride; op.pad = pad; op.depthwise = group != 1;
            op.Ho = (A.h + 2 * pad - k) / stride + 1; op.Wo = (A.w + 2 * pad - k) / stride + 1;
            op.kind = (k == 1 && group == 1 && stride == 1 && pad == 0) ? OP_PW : OP_KXK;
            op.in0 = in0;
            h->gmac += (double)op.Ho * op.Wo * wsize * 1e-9;
            const int oid = blob(L.outs[0]); Blob &ob = h->blobs[oid]; ob = Blob(); ob.c = outc; ob.h = op.Ho; ob.w = op.Wo; ob.n = (size_t)outc * op.Ho * op.Wo;
            op.out = oid; op.name = L.name;
            h->ops.push_back(op);
            continue;
        }
        if (L.type == "ReLU" || L.type == "Clip") {
            const int oid = blob(L.outs[0]);
            Op op; op.kind = OP_UNARY; op.in0 = in0; op.act = L.type == "ReLU" ? SGX_ACT_RELU : SGX_ACT_CLIP; op.lo = L.getf(0, 0.f); op.hi = L.getf(1, 0.f);
            Blob &ob = h->blobs[oid]; ob = A; ob.alias = -1; ob.d = nullptr;
            op.out = oid; op.name = L.name; h->ops.push_back(op); continue;
        }
        if (L.type == "BinaryOp") {
            const int in1 = resolve(blob(L.ins[1]));
            Op op; op.kind = OP_BINARY; op.in0 = in0; op.in1 = in1; op.bop = L.geti(0, 0);
            if (op.bop != 0 && op.bop != 2 && op.bop != 3) FAIL(SGX_ERR_UNSUPPORTED);
            if (!h->blobs[in1].scalar && h->blobs[in1].n != A.n) FAIL(SGX_ERR_UNSUPPORTED);
            const int oid = blob(L.outs[0]); Blob &ob = h->blobs[oid]; ob = A; ob.alias = -1; ob.d = nullptr;
            op.out = oid; op.name = L.name; h->ops.push_back(op); continue;
        }
        const int raw_in = blob(L.ins[0]);                                    // unresolved: keeps view layers (Permute) visible in the alias chain
        if (L.type == "Permute") { if (L.geti(0, 0) != 3) FAIL(SGX_ERR_UNSUPPORTED); const int oid = blob(L.outs[0]); h->blobs[oid] = A; h->blobs[oid].alias = raw_in; h->blobs[oid].sval = 1.f; /* marks "HWC view of" */ continue; }
        if (L.type == "Flatten") { const int oid = blob(L.outs[0]); h->blobs[oid] = A; h->blobs[oid].sval = 0.f; h->blobs[oid].alias = raw_in; continue; }
        if (L.type == "PriorBox") {
            const std::vector<float> mins = L.pa.count(0) ? L.pa.at(0) : std::vector<float>(), maxs = L.pa.count(1) ? L.pa.at(1) : std::vector<float>(), ars = L.pa.count(2) ? L.pa.at(2) : std::vector<float>();
            const int flip = L.geti(7, 1), clip = L.geti(8, 0); const float offset = L.getf(13, 0.f);
            const float step_w = (float)T / (float)A.w, step_h = (float)T / (float)A.h;
            const float var[4] = { L.getf(3, .1f), L.getf(4, .1f), L.getf(5, .2f), L.getf(6, .2f) };
            auto add = [&](float cx, float cy, float bw, float bh) {
                float bx[4] = { (cx - bw * 0.5f) / T, (cy - bh * 0.5f) / T, (cx + bw * 0.5f) / T, (cy + bh * 0.5f) / T };
                for (int q = 0; q < 4; q++) { prior_boxes.push_back(clip ? std::min(std::max(bx[q], 0.f), 1.f) : bx[q]); prior_vars.push_back(var[q]); }
            };
            for (int i = 0; i < A.h; i++) for (int j = 0; j < A.w; j++) {
                const float cx = ((float)j + offset) * step_w, cy = ((float)i + offset) * step_h;
                for (size_t kk = 0; kk < mins.size(); kk++) {
                    const float ms = mins[kk]; add(cx, cy, ms, ms);
                    if (!maxs.empty()) { const float s = sqrtf(ms * maxs[kk]); add(cx, cy, s, s); }
                    for (float ar : ars) { const float sq = sqrtf(ar); add(cx, cy, ms * sq, ms / sq); if (flip) add(cx, cy, ms / sq, ms * sq); }
                }
            }
            blob(L.outs[0]); continue;
        }
        if (L.type == "Concat") {
            if (L.name == "mbox_priorbox") { blob(L.outs[0]); continue; }
            size_t total = 0; for (const std::string &i : L.ins) total += h->blobs[resolve(blob(i))].n;
            const int oid = blob(L.outs[0]); Blob &ob = h->blobs[oid]; ob.c = 1; ob.h = 1; ob.w = (int)total; ob.n = total;
            int off = 0;
            for (const std::string &i : L.ins) {
                // input is Flatten(Permute(conv)) : find whether a Permute sits in the alias chain
                int id = blob(i); bool hwc = false; while (h->blobs[id].alias >= 0) { if (h->blobs[id].sval == 1.f) hwc = true; id = h->blobs[id].alias; }
                Op op; op.kind = hwc ? OP_PERMUTE_INTO : OP_COPY_INTO; op.in0 = id; op.out = oid; op.off = off; op.C = h->blobs[id].c; op.rows = h->blobs[id].h * h->blobs[id].w;
                op.name = L.name; h->ops.push_back(op); off += (int)h->blobs[id].n;
            }
            continue;
        }
        if (L.type == "Reshape") { const int oid = blob(L.outs[0]); h->blobs[oid] = A; h->blobs[oid].alias = raw_in; h->blobs[oid].w = L.geti(0, 1); h->blobs[oid].h = (int)(A.n / L.geti(0, 1)); continue; }
        if (L.type == "Softmax") {
            Op op; op.kind = OP_SOFTMAX; op.in0 = in0; op.C = h->blobs[blob(L.ins[0])].w; op.rows = (int)(A.n / op.C);
            if (op.C < 1 || op.C > SGX_SOFTMAX_MAXC) FAIL(SGX_ERR_UNSUPPORTED);
            const int oid = blob(L.outs[0]); Blob &ob = h->blobs[oid]; ob = A; ob.alias = -1; ob.d = nullptr;
            op.out = oid; op.name = L.name; h->ops.push_back(op); continue;
        }
        if (L.type == "DetectionOutput") {
            h->loc_blob = resolve(blob(L.ins[0])); h->conf_blob = resolve(blob(L.ins[1]));
            h->num_class = L.geti(0, 21); h->nms_th = L.getf(1, 0.45f); h->nms_top_k = L.geti(2, 300); h->keep_top_k = L.geti(3, 100); h->conf_th = L.getf(4, 0.01f);
            h->dvar[0] = L.getf(5, .1f); h->dvar[1] = L.getf(6, .1f); h->dvar[2] = L.getf(7, .2f); h->dvar[3] = L.getf(8, .2f);
            continue;
        }
        FAIL(SGX_ERR_UNSUPPORTED);
    }
    if (h->loc_blob < 0 || h->conf_blob < 0) FAIL(SGX_ERR_INVALID);
    // ---- fusion pass: fold the elementwise chain behind every convolution into its epilogue, and the Permute+Flatten+Concat copy of
    // the head convolutions into an HWC store.  Same fp32 operations in the same order: the fused plan is bit-identical to the unfused one.
    if (g_det_fuse) {
        std::vector<Op> &ops = h->ops;
        const int nops = (int)ops.size();
        std::vector<int> producer(h->blobs.size(), -2);
        producer[resolve(h->blob_id.at("input"))] = -1;
        for (int i = 0; i < nops; i++) if (ops[i].kind != OP_PERMUTE_INTO && ops[i].kind != OP_COPY_INTO) producer[ops[i].out] = i;
        auto readers = [&](int id, std::vector<int> &r) {
            r.clear();
            for (int i = 0; i < nops; i++) {
                if (ops[i].dead) continue;
                const bool r0 = ops[i].in0 == id, r1 = ops[i].kind == OP_BINARY && ops[i].in1 == id && !h->blobs[id].scalar;
                if (r0 || r1) r.push_back(i);
            }
        };
        std::vector<int> R;
        for (int ci = 0; ci < nops; ci++) {
            Op &cv = ops[ci];
            if (cv.kind != OP_PW && cv.kind != OP_KXK) continue;
            const int root = cv.out;
            int cur = root, root_extra = -1;
            std::vector<EpiStep> steps; std::vector<int> absorbed;
            while ((int)steps.size() < SGX_EPI_MAX) {
                if (cur == h->loc_blob || cur == h->conf_blob) break;
                readers(cur, R);
                int next = -1;
                if (R.size() == 1) next = R[0];
                else if (cur == root && R.size() == 2 && steps.empty()) { next = std::min(R[0], R[1]); root_extra = std::max(R[0], R[1]); }
                else break;
                if (next == root_extra && cur != root) { /* handled below as the ROOT operand */ }
                const Op &e = ops[next];
                EpiStep st; st.a = 0; st.b = 0; st.tensor = -1; st.src = SGX_ESRC_CONST;
                if (e.kind == OP_UNARY) { st.op = e.act == SGX_ACT_RELU ? SGX_EOP_RELU : SGX_EOP_CLIP; st.a = e.lo; st.b = e.hi; }
                else if (e.kind == OP_BINARY) {
                    if (e.in0 == cur && e.in1 == cur) break;
                    const bool first = e.in0 == cur;                     // chain value is the left operand
                    const int other = first ? e.in1 : e.in0;
                    const int fwd = e.bop == 0 ? SGX_EOP_ADD : e.bop == 2 ? SGX_EOP_MUL : SGX_EOP_DIV;
                    st.op = first ? fwd : (e.bop == 3 ? SGX_EOP_RDIV : fwd);
                    if (h->blobs[other].scalar) { st.src = SGX_ESRC_CONST; st.a = h->blobs[other].sval; }
                    else if (other == root && cur != root) { st.src = SGX_ESRC_ROOT; }
                    else if (producer[other] >= -1 && producer[other] < ci && h->blobs[other].n == h->blobs[root].n) { st.src = SGX_ESRC_TENSOR; st.tensor = other; }
                    else break;
                } else break;
                steps.push_back(st); absorbed.push_back(next); cur = e.out;
            }
            // the convolution's raw output stays in registers: every reader of it must have been absorbed
            if (root_extra >= 0 && std::find(absorbed.begin(), absorbed.end(), root_extra) == absorbed.end()) { steps.clear(); absorbed.clear(); cur = root; }
            // ... and a ROOT operand may only be used when the raw output has no reader left outside the chain
            if (!steps.empty()) {
                cv.epi = steps; cv.out = cur;
                for (int a : absorbed) ops[a].dead = true;
                producer[cur] = ci;
            }
            if (cv.kind == OP_PW) {                                      // head convolution -> HWC store into the concat buffer
                bool tens = false; for (const EpiStep &st : cv.epi) tens = tens || st.src == SGX_ESRC_TENSOR;
                readers(cv.out, R);
                if (!tens && R.size() == 1 && ops[R[0]].kind == OP_PERMUTE_INTO && cv.out != h->loc_blob && cv.out != h->conf_blob) {
                    cv.hwc = 1; cv.hwc_off = ops[R[0]].off; cv.out = ops[R[0]].out; ops[R[0]].dead = true;
                }
            }
        }
        // ---- block fusion: pointwise expand (ReLU / Clip) -> depthwise (ReLU / Clip) -> pointwise project (nothing / + tensor) with no other reader of the two
        // intermediates becomes one OP_FUSED_BLOCK (sgx_det_block.h): the expanded tensor never reaches HBM.  OPT-IN (SGX_DET_BLOCK_FUSION=1 or
        // sgx_det_debug_set_block_fusion): bit-identical, but measured SLOWER than the three tuned kernels on MI355X at batch 256 (13.7 vs 10.9 ms per forward: 30-60 k small
        // workgroups, each re-staging its weights and running five barrier-separated phases at 3 waves per SIMD) — see DESIGN.md §6.
        // k_fused_block2 (VALU-only, thread per pixel) takes the high-resolution few-channel blocks by default (SGX_DET_BLOCK2=0 turns it off); faster than the three kernels there.
        static const int fb2_env = sgx_getenv("SGX_DET_BLOCK2") ? atoi(sgx_getenv("SGX_DET_BLOCK2")) : 1;
        const bool fb2_on = fb2_env != 0 && !g_det_legacy && g_det_fuse;
        static const int hrb_env = sgx_getenv("SGX_DET_HRB") ? atoi(sgx_getenv("SGX_DET_HRB")) : 1;
        const bool hrb_on = hrb_env != 0;
#ifdef SGX_DEBUG_TAPS
        const bool fb1_on = (g_det_block_fusion || sgx_getenv("SGX_DET_BLOCK_FUSION")) && !g_det_legacy;
#else
        const bool fb1_on = false;                                  // k_fused_block exists in the tap build only
#endif
        if (fb1_on || fb2_on) {
            auto act_only = [&](const Op &o, float *lo, float *hi) -> bool {
                if (o.epi.size() != 1) return false;
                const EpiStep &st = o.epi[0];
                if (st.op == SGX_EOP_RELU) { *lo = 0.f; *hi = INFINITY; return true; }
                if (st.op == SGX_EOP_CLIP) { *lo = st.a; *hi = st.b; return true; }
                return false;
            };
            for (int ai = 0; ai < nops; ai++) {
                Op &a = ops[ai];
                if (a.dead || a.kind != OP_PW || a.hwc) continue;
                float lo1, hi1, lo2, hi2;
                if (!act_only(a, &lo1, &hi1)) continue;
                readers(a.out, R); if (R.size() != 1) continue;
                const int bi = R[0]; Op &bq = ops[bi];
                if (bq.kind != OP_KXK || !bq.depthwise || (bq.k != 3 && bq.k != 5) || !act_only(bq, &lo2, &hi2)) continue;
                readers(bq.out, R); if (R.size() != 1) continue;
                const int ci = R[0]; Op &c = ops[ci];
                if (c.kind != OP_PW || c.hwc || c.in0 != bq.out) continue;
                int res = -1;
                if (c.epi.size() == 1 && c.epi[0].op == SGX_EOP_ADD && c.epi[0].src == SGX_ESRC_TENSOR) res = c.epi[0].tensor;
                else if (!c.epi.empty()) continue;
                if (a.outc != bq.outc || c.inc != bq.outc) continue;
                if (a.out == h->loc_blob || a.out == h->conf_blob || bq.out == h->loc_blob || bq.out == h->conf_blob) continue;
                const int v2 = sgx_fb2_variant(a.inc, c.outc, bq.k, bq.stride);
                if (fb2_on && v2 && (a.outc % sgx_fb2_cm(v2)) == 0 && c.wtT && bq.pad == bq.k / 2) {
                    SgxFusedBlk fb; memset(&fb, 0, sizeof fb);
                    fb.Cin = a.inc; fb.Cmid = a.outc; fb.Cout = c.outc; fb.K = bq.k; fb.stride = bq.stride; fb.pad = bq.pad; fb.H = a.H; fb.W = a.W; fb.Ho = bq.Ho; fb.Wo = bq.Wo;
                    fb.lo1 = lo1; fb.hi1 = hi1; fb.lo2 = lo2; fb.hi2 = hi2; fb.v2 = v2;
                    sgx_fb2_tile(v2, &fb.TOH, &fb.TOW);
                    fb.tiles_x = (fb.Wo + fb.TOW - 1) / fb.TOW; fb.tiles_y = (fb.Ho + fb.TOH - 1) / fb.TOH;
                    fb.w1 = a.wt; fb.b1 = a.bias; fb.wd = bq.wt; fb.bd = bq.bias; fb.w2 = c.wt; fb.b2 = c.bias; fb.w2t = c.wtT; fb.ldw2 = c.ldw;
                    {   // depthwise weights with the channels of a pair interleaved
                        const int kk = bq.k * bq.k; std::vector<float> wh((size_t)fb.Cmid * kk), w2((size_t)fb.Cmid * kk);
                        if (hipMemcpy(wh.data(), bq.wt, wh.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                        for (int m = 0; m < fb.Cmid; m++) for (int t = 0; t < kk; t++) w2[((size_t)(m >> 1) * kk + t) * 2 + (m & 1)] = wh[(size_t)m * kk + t];
                        float *dw2 = nullptr; if (h->alloc(&dw2, w2.size())) FAIL(SGX_ERR_NOMEM);
                        if (hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                        fb.wd2 = dw2;
                    }
                    // round 6, bf16x3 plan: the same block with both pointwise convolutions on the bf16 matrix pipes (k_hrb, sgx_det_hrb.h); SGX_DET_HRB=0 (tap) keeps k_fused_block2
                    if (hrb_on && h->gemm == 1 && a.wS && c.wS && sgx_hrb_variant(fb.Cin, fb.Cmid, fb.Cout, fb.K, fb.stride, 0, res >= 0, fb.lo1, fb.lo2, &fb.TOH, &fb.TOW, &fb.hrb_occ)) {
                        fb.hrb = 1; fb.w1S = a.wS; fb.ld1S = a.ldw; fb.w2S = c.wS; fb.ld2S = c.ldw;
                        fb.tiles_x = (fb.Wo + fb.TOW - 1) / fb.TOW; fb.tiles_y = (fb.Ho + fb.TOH - 1) / fb.TOH;
                    }
                    Op f; f.kind = OP_FUSED_BLOCK; f.in0 = a.in0; f.out = c.out; f.name = a.name + "+" + bq.name + "+" + c.name; f.fb = fb; f.fb_res_blob = res;
                    f.inc = a.inc; f.outc = c.outc; f.H = a.H; f.W = a.W; f.Ho = bq.Ho; f.Wo = bq.Wo; f.k = bq.k; f.stride = bq.stride;
                    ops[ci] = f; a.dead = true; bq.dead = true;
                    continue;
                }
                if (!fb1_on) continue;
                if ((a.inc & 1) || (a.outc & 1) || a.inc > 64 || c.outc > 64) continue;       // k_fused_block stages <= 2048 weights per array and chunk
                // tile choice: least matrix-core work per output pixel among the tiles that fit the LDS budget and the accumulator registers
                SgxFusedBlk fb; memset(&fb, 0, sizeof fb);
                fb.Cin = a.inc; fb.Cmid = a.outc; fb.Cout = c.outc; fb.K = bq.k; fb.stride = bq.stride; fb.pad = bq.pad; fb.H = a.H; fb.W = a.W; fb.Ho = bq.Ho; fb.Wo = bq.Wo;
                fb.lo1 = lo1; fb.hi1 = hi1; fb.lo2 = lo2; fb.hi2 = hi2;
                const int ncb = (fb.Cout + 31) / 32, nch = (fb.Cmid + 31) / 32;
                double best = 1e30; int bth = 0, btw = 0;
                for (int th = 1; th <= 16; th++) for (int tw = 4; tw <= 40; tw++) {
                    SgxFusedBlk t = fb; t.TOH = th; t.TOW = tw; t.TIH = (th - 1) * fb.stride + fb.K; t.TIW = (tw - 1) * fb.stride + fb.K;
                    t.NPI = ((t.TIH * t.TIW + 31) / 32) * 32; t.NPO = ((th * tw + 31) / 32) * 32; t.CMR = std::min(32, fb.Cmid); t.ES = t.NPI + 4;
                    if (ncb * (t.NPO / 32) > 8 || sgx_fb_lds_floats(t) * 4 > 52 * 1024 || fb.Cout * th * tw > 64 * 1024 || t.NPI >= (1 << 12) || th * tw >= (1 << 12)) continue;
                    const int txn = (fb.Wo + tw - 1) / tw, tyn = (fb.Ho + th - 1) / th;
                    // MFMAs per tile: expand NBI * Cin/2 per chunk, project NBO * ncb * 16 per chunk; VALU depthwise ~ K*K per output per channel (weighted); fixed per-tile cost
                    const double cost = (double)txn * tyn * (nch * ((t.NPI / 32) * (fb.Cin / 2.0) + (t.NPO / 32) * ncb * 16.0) * 64.0 / 4.0 + (double)fb.Cmid * th * tw * fb.K * fb.K * 4.0 / 256.0 * 2.0 + 3000.0);
                    if (cost < best) { best = cost; bth = th; btw = tw; }
                }
                if (!bth) continue;
                fb.TOH = bth; fb.TOW = btw; fb.TIH = (bth - 1) * fb.stride + fb.K; fb.TIW = (btw - 1) * fb.stride + fb.K;
                fb.NPI = ((fb.TIH * fb.TIW + 31) / 32) * 32; fb.NPO = ((bth * btw + 31) / 32) * 32; fb.CMR = std::min(32, fb.Cmid); fb.ES = fb.NPI + 4;
                { auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
                  fb.m_tiw = magic(fb.TIW); fb.m_tow = magic(btw); fb.m_npo = magic(bth * btw); fb.m_kk = magic(fb.K * fb.K); }
                fb.tiles_x = (fb.Wo + btw - 1) / btw; fb.tiles_y = (fb.Ho + bth - 1) / bth; fb.dbg = sgx_getenv("SGX_FB_DBG") ? atoi(sgx_getenv("SGX_FB_DBG")) : 0;
                fb.w1 = a.wt; fb.b1 = a.bias; fb.wd = bq.wt; fb.bd = bq.bias; fb.w2 = c.wt; fb.b2 = c.bias;
                Op f; f.kind = OP_FUSED_BLOCK; f.in0 = a.in0; f.out = c.out; f.name = a.name + "+" + bq.name + "+" + c.name; f.fb = fb; f.fb_res_blob = res;
                f.inc = a.inc; f.outc = c.outc; f.H = a.H; f.W = a.W; f.Ho = bq.Ho; f.Wo = bq.Wo; f.k = bq.k; f.stride = bq.stride;
                ops[ci] = f; a.dead = true; bq.dead = true;           // the block takes the project convolution's place in the plan (its residual operand is older)
            }
        }
        // ---- inverted-residual blocks on the matrix cores (sgx_det_irb.h): [pointwise expand + act ->] depthwise + act -> pointwise project
        // [-> squeeze (ReLU) -> excite -> hard-sigmoid gate x project output] [+ residual], one kernel, nothing but the block's input and output in HBM.
        // SGX_DET_IRB=0 keeps the per-layer plan (bit-identical either way).
        static const int irb_env = sgx_getenv("SGX_DET_IRB") ? atoi(sgx_getenv("SGX_DET_IRB")) : 1;
        const int irb_mode = g_det_irb < 0 ? irb_env : g_det_irb;
        if (irb_mode != 0 && !g_det_legacy) {
            struct EpiClass { int mode; float c1, lo, hi, c2; int t0, t1; };
            auto classify = [&](const std::vector<EpiStep> &e) -> EpiClass {
                EpiClass r; r.mode = SGX_EMODE_GENERIC; r.c1 = 0; r.lo = 0; r.hi = INFINITY; r.c2 = 1; r.t0 = -1; r.t1 = -1;
                auto is = [&](size_t i, int opc, int src) { return i < e.size() && e[i].op == opc && (opc == SGX_EOP_CLIP || opc == SGX_EOP_RELU || e[i].src == src); };
                if (e.empty()) r.mode = SGX_EMODE_NONE;
                else if (e.size() == 1 && is(0, SGX_EOP_RELU, 0)) { r.mode = SGX_EMODE_ACT; r.lo = 0.f; r.hi = INFINITY; }
                else if (e.size() == 1 && is(0, SGX_EOP_CLIP, 0)) { r.mode = SGX_EMODE_ACT; r.lo = e[0].a; r.hi = e[0].b; }
                else if (e.size() == 1 && is(0, SGX_EOP_ADD, SGX_ESRC_TENSOR)) { r.mode = SGX_EMODE_ADD_T; r.t1 = e[0].tensor; }
                else if (e.size() == 4 && is(0, SGX_EOP_ADD, SGX_ESRC_CONST) && is(1, SGX_EOP_CLIP, 0) && is(2, SGX_EOP_MUL, SGX_ESRC_ROOT) && is(3, SGX_EOP_DIV, SGX_ESRC_CONST)) {
                    r.mode = SGX_EMODE_HSWISH; r.c1 = e[0].a; r.lo = e[1].a; r.hi = e[1].b; r.c2 = e[3].a;
                } else if ((e.size() == 4 || e.size() == 5) && is(0, SGX_EOP_ADD, SGX_ESRC_CONST) && is(1, SGX_EOP_CLIP, 0) && is(2, SGX_EOP_DIV, SGX_ESRC_CONST) && is(3, SGX_EOP_MUL, SGX_ESRC_TENSOR) &&
                           (e.size() == 4 || is(4, SGX_EOP_ADD, SGX_ESRC_TENSOR))) {
                    r.mode = e.size() == 4 ? SGX_EMODE_GATE : SGX_EMODE_GATE_ADD; r.c1 = e[0].a; r.lo = e[1].a; r.hi = e[1].b; r.c2 = e[2].a; r.t0 = e[3].tensor; if (e.size() == 5) r.t1 = e[4].tensor;
                }
                return r;
            };
            // readers of a blob among the live ops, INCLUDING epilogue tensor operands
            auto readers_all = [&](int id, std::vector<int> &r) {
                r.clear();
                for (int i = 0; i < nops; i++) {
                    if (ops[i].dead) continue;
                    bool rd = ops[i].in0 == id || (ops[i].kind == OP_BINARY && ops[i].in1 == id && !h->blobs[id].scalar) || (ops[i].kind == OP_FUSED_BLOCK && ops[i].fb_res_blob == id) ||
                              (ops[i].kind == OP_IRB && ops[i].irb_res_blob == id) || (ops[i].kind == OP_SE_GATE && ops[i].sg_res_blob == id);
                    for (const EpiStep &st : ops[i].epi) rd = rd || st.tensor == id;
                    if (rd) r.push_back(i);
                }
            };
            static const int irb_mask = sgx_getenv("SGX_DET_IRB_MASK") ? atoi(sgx_getenv("SGX_DET_IRB_MASK")) : 0xff;      // tuning tap: 1 stride-1 blocks, 2 stride-2 blocks, 4 no-expand blocks, 8 heads
            for (int bi = 0; bi < nops; bi++) {
                Op &bq = ops[bi];
                if (bq.dead || bq.kind != OP_KXK || !bq.depthwise || (bq.k != 3 && bq.k != 5) || (bq.stride != 1 && bq.stride != 2) || bq.pad != bq.k / 2 || bq.inc != bq.outc) continue;
                const EpiClass cb = classify(bq.epi);
                if (cb.mode != SGX_EMODE_ACT && cb.mode != SGX_EMODE_HSWISH) continue;
                if (bq.out == h->loc_blob || bq.out == h->conf_blob) continue;
                // project: the only reader of the depthwise output
                readers_all(bq.out, R); if (R.size() != 1) continue;
                const int ci = R[0]; Op &c = ops[ci];
                if (c.kind != OP_PW || c.in0 != bq.out || !c.wtT || c.inc != bq.outc || (c.inc & 1)) continue;
                const EpiClass cc = classify(c.epi);
                if (cc.mode != SGX_EMODE_NONE && cc.mode != SGX_EMODE_ADD_T) continue;
                if (c.hwc && cc.mode != SGX_EMODE_NONE) continue;
                // expand: pointwise producer of the depthwise input whose only reader is the depthwise convolution
                int ai = -1; EpiClass ca = { SGX_EMODE_NONE, 0.f, 0.f, INFINITY, 1.f, -1, -1 };
                for (int i = 0; i < nops; i++) if (!ops[i].dead && ops[i].kind == OP_PW && ops[i].out == bq.in0 && !ops[i].hwc) ai = i;
                if (ai >= 0) {
                    readers_all(bq.in0, R);
                    ca = classify(ops[ai].epi);
                    if (R.size() != 1 || !ops[ai].wtT || (ops[ai].inc & 1) || (ca.mode != SGX_EMODE_ACT && ca.mode != SGX_EMODE_HSWISH) || bq.in0 == h->loc_blob || bq.in0 == h->conf_blob) ai = -1;
                }
                // squeeze-excite behind the project convolution: readers of its output = { squeeze conv, excite conv's gate operand }
                int di = -1, ei = -1; EpiClass cd = { SGX_EMODE_NONE, 0.f, 0.f, INFINITY, 1.f, -1, -1 }, ce = cd; int out_blob = c.out, res_blob = cc.mode == SGX_EMODE_ADD_T ? cc.t1 : -1;
                if (!c.hwc && cc.mode == SGX_EMODE_NONE && c.out != h->loc_blob && c.out != h->conf_blob) {
                    readers_all(c.out, R);
                    if (R.size() == 2) {
                        for (int q = 0; q < 2; q++) {
                            const Op &d = ops[R[q]], &e = ops[R[1 - q]];
                            if (d.kind != OP_PW || e.kind != OP_PW || d.in0 != c.out || d.hwc || e.hwc || !d.wtT || !e.wtT || e.in0 != d.out || (d.outc & 1) || (d.inc & 1)) continue;
                            cd = classify(d.epi); ce = classify(e.epi);
                            if (cd.mode != SGX_EMODE_ACT || (ce.mode != SGX_EMODE_GATE && ce.mode != SGX_EMODE_GATE_ADD) || ce.t0 != c.out || e.outc != c.outc) continue;
                            std::vector<int> R2; readers_all(d.out, R2); if (R2.size() != 1) continue;
                            di = R[q]; ei = R[1 - q]; out_blob = e.out; res_blob = ce.mode == SGX_EMODE_GATE_ADD ? ce.t1 : -1;
                        }
                    }
                }
                // the 75 -> 38 block (24 -> 72 -> 40, 5 x 5 depthwise stride 2, squeeze-excite tail): a k loop of 24 is too short for the matrix-core kernel and the per-layer plan
                // moves the 72-channel expansion through HBM twice; k_fused_block2 with the squeeze-excite tail in registers takes it
                // (SGX_DET_BLOCK2_SE=0: per-layer kernels; irb mode 2 keeps k_irb on these shapes for its tests)
                static const int fb2se_env = sgx_getenv("SGX_DET_BLOCK2_SE") ? atoi(sgx_getenv("SGX_DET_BLOCK2_SE")) : 1;
                if (fb2_on && fb2se_env && irb_mode == 1 && ai >= 0 && di >= 0 && !c.hwc && ca.mode == SGX_EMODE_ACT && cb.mode == SGX_EMODE_ACT) {
                    Op &a = ops[ai]; const Op &d = ops[di], &e = ops[ei];
                    const int v2 = sgx_fb2_variant(a.inc, c.outc, bq.k, bq.stride, d.outc);
                    // round 6: k_hrb also takes the two 40 -> 120 -> 40 blocks at 38 x 38 (5 x 5, squeeze-excite, + residual) that k_fused_block2 lost to the per-layer kernels
                    int h_toh = 0, h_tow = 0, h_occ = 0;
                    const bool hrb_ok = hrb_on && h->gemm == 1 && a.wS && c.wS && d.wS && e.wS && bq.pad == bq.k / 2 &&
                                        sgx_hrb_variant(a.inc, a.outc, c.outc, bq.k, bq.stride, d.outc, res_blob >= 0, ca.lo, cb.lo, &h_toh, &h_tow, &h_occ);
                    if ((hrb_ok || (v2 && (a.outc % sgx_fb2_cm(v2)) == 0)) && a.outc == bq.outc && d.inc == c.outc && e.inc == d.outc) {
                        SgxFusedBlk fb; memset(&fb, 0, sizeof fb);
                        fb.Cin = a.inc; fb.Cmid = a.outc; fb.Cout = c.outc; fb.K = bq.k; fb.stride = bq.stride; fb.pad = bq.pad; fb.H = a.H; fb.W = a.W; fb.Ho = bq.Ho; fb.Wo = bq.Wo;
                        fb.lo1 = ca.lo; fb.hi1 = ca.hi; fb.lo2 = cb.lo; fb.hi2 = cb.hi; fb.v2 = v2;
                        if (v2) sgx_fb2_tile(v2, &fb.TOH, &fb.TOW); else { fb.TOH = h_toh; fb.TOW = h_tow; }
                        fb.tiles_x = (fb.Wo + fb.TOW - 1) / fb.TOW; fb.tiles_y = (fb.Ho + fb.TOH - 1) / fb.TOH;
                        fb.w1 = a.wt; fb.b1 = a.bias; fb.wd = bq.wt; fb.bd = bq.bias; fb.w2 = c.wt; fb.b2 = c.bias; fb.w2t = c.wtT; fb.ldw2 = c.ldw;
                        fb.Cq = d.outc; fb.qlo = cd.lo; fb.qhi = cd.hi; fb.gc1 = ce.c1; fb.glo = ce.lo; fb.ghi = ce.hi; fb.gc2 = ce.c2;
                        fb.wq1 = d.wt; fb.bq1 = d.bias; fb.wq2 = e.wt; fb.bq2 = e.bias;
                        {   // depthwise weights with the channels of a pair interleaved; squeeze / excite weights with the output pairs interleaved
                            const int kk = bq.k * bq.k, Cq = d.outc, Co = c.outc;
                            std::vector<float> wh((size_t)fb.Cmid * kk), w2((size_t)fb.Cmid * kk), q1((size_t)Cq * Co), q2((size_t)Co * Cq), p1((size_t)Cq * Co), p2((size_t)Co * Cq);
                            if (hipMemcpy(wh.data(), bq.wt, wh.size() * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(q1.data(), d.wt, q1.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                                hipMemcpy(q2.data(), e.wt, q2.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                            for (int m = 0; m < fb.Cmid; m++) for (int t = 0; t < kk; t++) w2[((size_t)(m >> 1) * kk + t) * 2 + (m & 1)] = wh[(size_t)m * kk + t];
                            for (int j = 0; j < Cq; j++) for (int k = 0; k < Co; k++) p1[((size_t)(j >> 1) * Co + k) * 2 + (j & 1)] = q1[(size_t)j * Co + k];
                            for (int co = 0; co < Co; co++) for (int j = 0; j < Cq; j++) p2[((size_t)j * (Co / 2) + (co >> 1)) * 2 + (co & 1)] = q2[(size_t)co * Cq + j];
                            float *dw2 = nullptr, *dp1 = nullptr, *dp2 = nullptr;
                            if (h->alloc(&dw2, w2.size()) || h->alloc(&dp1, p1.size()) || h->alloc(&dp2, p2.size())) FAIL(SGX_ERR_NOMEM);
                            if (hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dp1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                                hipMemcpy(dp2, p2.data(), p2.size() * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                            fb.wd2 = dw2; fb.wq1p = dp1; fb.wq2p = dp2;
                        }
                        if (hrb_on && h->gemm == 1 && a.wS && c.wS && d.wS && e.wS && sgx_hrb_variant(fb.Cin, fb.Cmid, fb.Cout, fb.K, fb.stride, fb.Cq, res_blob >= 0, fb.lo1, fb.lo2, &fb.TOH, &fb.TOW, &fb.hrb_occ)) {
                            fb.hrb = 1; fb.w1S = a.wS; fb.ld1S = a.ldw; fb.w2S = c.wS; fb.ld2S = c.ldw; fb.wq1S = d.wS; fb.ldq1S = d.ldw; fb.wq2S = e.wS; fb.ldq2S = e.ldw;
                            fb.tiles_x = (fb.Wo + fb.TOW - 1) / fb.TOW; fb.tiles_y = (fb.Ho + fb.TOH - 1) / fb.TOH;
                        }
                        Op f; f.kind = OP_FUSED_BLOCK; f.in0 = a.in0; f.out = out_blob; f.fb = fb; f.fb_res_blob = res_blob;
                        f.name = a.name + "+" + bq.name + "+" + c.name + "+" + d.name + "+" + e.name;
                        f.inc = a.inc; f.outc = c.outc; f.H = a.H; f.W = a.W; f.Ho = bq.Ho; f.Wo = bq.Wo; f.k = bq.k; f.stride = bq.stride;
                        ops[ai].dead = true; ops[bi].dead = true; ops[ci].dead = true; ops[di].dead = true; ops[ei] = f;      // the block takes the place of its LAST convolution in the plan
                        continue;
                    }
                }
                const bool head = c.hwc != 0;
                const int kind_bit = head ? 8 : (ai < 0 ? 4 : (bq.stride == 2 ? 2 : 1));
                if (!(irb_mask & kind_bit)) continue;
                if (irb_mode == 1) {
                    // measured at 512 frames (profiles/r3_detector_ops.txt against profiles/r2_detector_ops.txt): the kernel wins on the 19 x 19 blocks, on the 38 -> 19
                    // stride-2 block and on every SSD head; the high-resolution few-channel blocks (short k loops), the stride-2 block without an expand stage and
                    // the 10 x 10 block stay on the per-layer kernels
                    const int px = bq.Ho * bq.Wo;
                    const bool win = head || (ai >= 0 && ops[ai].inc >= 40 && px >= 256 && px <= 400);
                    if (!win) continue;
                }
                const int NT = (c.outc + 31) / 32, NQ = di >= 0 ? (ops[di].outc + 31) / 32 : 0;
                if (!sgx_irb_supported(bq.k, bq.stride, NT, NQ, ai >= 0, cb.mode == SGX_EMODE_HSWISH)) continue;
                if (ai >= 0 && ca.mode != cb.mode) continue;                      // one activation kind per instantiation
                if (bq.outc % 8) continue;                                        // stage B advances four k-steps per trip
                // geometry: whole images (G per workgroup) or bands of output rows; two plane buffers when they fit
                SgxIrb ib; memset(&ib, 0, sizeof ib);
                ib.Cin = ai >= 0 ? ops[ai].inc : bq.inc; ib.Cexp = bq.outc; ib.Cout = c.outc; ib.Cq = di >= 0 ? ops[di].outc : 0;
                ib.H = bq.H; ib.W = bq.W; ib.Ho = bq.Ho; ib.Wo = bq.Wo; ib.K = bq.k; ib.S = bq.stride; ib.pad = bq.pad;
                ib.Wp = bq.W + 2 * bq.pad;
                const size_t lds_max = 160 * 1024; const int kkp = SGX_IRB_KKP(bq.k), max_px = 384;
                auto lds_of = [&](int g, int oh, int nb) { const int planeT = ((g * ((oh - 1) * bq.stride + bq.k) * ib.Wp + 3) / 4) * 4; return (size_t)nb * 32 * (planeT + kkp) * 4; };
                int G = 0, OH = bq.Ho, nbands = 1, nbuf = 2;
                static const int env_g = sgx_getenv("SGX_IRB_G") ? atoi(sgx_getenv("SGX_IRB_G")) : 0, env_nbuf = sgx_getenv("SGX_IRB_NBUF") ? atoi(sgx_getenv("SGX_IRB_NBUF")) : 0,
                                 env_split = sgx_getenv("SGX_IRB_SPLIT") ? atoi(sgx_getenv("SGX_IRB_SPLIT")) : 0, env_minhw = sgx_getenv("SGX_IRB_MINHW") ? atoi(sgx_getenv("SGX_IRB_MINHW")) : 0;      // tuning taps
                if (bq.Ho * bq.Wo < env_minhw) continue;
                const bool split = env_split > 0 && bq.Ho * bq.Wo >= env_split;          // force bands of about half the image
                if (!split && bq.Ho * bq.Wo <= max_px && lds_of(1, bq.Ho, 1) <= lds_max) {
                    // one image per workgroup (two of the tiny maps): many small workgroups fill the chip and overlap each other's barriers better than a few
                    // large ones (measured at 512 frames: 10 x 10 maps 0.91 -> 0.78 ms, 5 x 5 heads 0.15 -> 0.08 ms against three / twelve images per workgroup)
                    G = std::min(std::max(1, max_px / (bq.Ho * bq.Wo)), bq.Ho * bq.Wo >= 64 ? 1 : 2);
                    if (env_g > 0) G = std::min(std::max(1, max_px / (bq.Ho * bq.Wo)), env_g);
                    while (G > 1 && lds_of(G, bq.Ho, 2) > lds_max) G--;
                    nbuf = lds_of(G, bq.Ho, 2) <= lds_max ? 2 : 1;
                } else {
                    G = 1; OH = std::max(1, std::min(split ? (bq.Ho + 1) / 2 : bq.Ho, max_px / bq.Wo));
                    while (OH > 1 && lds_of(1, OH, 1) > lds_max) OH--;
                    if (lds_of(1, OH, 1) > lds_max || OH * bq.Wo > 1024) continue;
                    nbands = (bq.Ho + OH - 1) / OH; OH = (bq.Ho + nbands - 1) / nbands;              // even bands
                    nbuf = lds_of(1, OH, 2) <= lds_max ? 2 : 1;
                }
                if (env_nbuf == 1) nbuf = 1;
                ib.G = G; ib.OH = OH; ib.nbands = nbands; ib.nbuf = nbuf;
                ib.HpWp = ((OH - 1) * bq.stride + bq.k) * ib.Wp; ib.planeT = ((G * ib.HpWp + 3) / 4) * 4;
                {   // 32-bit offsets inside the kernel
                    const size_t big = (size_t)B * std::max(std::max(h->blobs[ai >= 0 ? ops[ai].in0 : bq.in0].n, h->blobs[out_blob].n), (size_t)1) * 4;
                    if (big >= 0xFFFFFFFFull) continue;
                }
                ib.has_expand = ai >= 0;
                { static const int env_stagger = sgx_getenv("SGX_IRB_STAGGER") ? atoi(sgx_getenv("SGX_IRB_STAGGER")) : 1; ib.stagger = env_stagger; }
                if (ai >= 0) { ib.act1 = ca.mode; ib.a1c1 = ca.c1; ib.a1lo = ca.lo; ib.a1hi = ca.hi; ib.a1c2 = ca.c2; ib.w1T = ops[ai].wtT; ib.b1 = ops[ai].bias; ib.ld1 = ops[ai].ldw; ib.w1 = ops[ai].wt; }
                ib.act2 = cb.mode; ib.a2c1 = cb.c1; ib.a2lo = cb.lo; ib.a2hi = cb.hi; ib.a2c2 = cb.c2;
                ib.w2T = c.wtT; ib.b2 = c.bias; ib.ld2 = c.ldw; ib.w2 = c.wt; ib.wd = bq.wt; ib.bd = bq.bias;
                if (di >= 0) {
                    ib.qlo = cd.lo; ib.qhi = cd.hi; ib.gc1 = ce.c1; ib.glo = ce.lo; ib.ghi = ce.hi; ib.gc2 = ce.c2;
                    ib.wq1T = ops[di].wtT; ib.bq1 = ops[di].bias; ib.ldq1 = ops[di].ldw; ib.wq1 = ops[di].wt;
                    ib.wq2T = ops[ei].wtT; ib.bq2 = ops[ei].bias; ib.ldq2 = ops[ei].ldw; ib.wq2 = ops[ei].wt;
                }
                ib.has_res = res_blob >= 0; ib.hwc = c.hwc; ib.hwc_off = c.hwc_off;
                {   // tuning tap SGX_IRB_W1LDS=1: expand weights through LDS when the two slices fit beside the planes (the 5 x 5 blocks' planes leave no room).  OFF: measured 6 % slower
                    // on every block (3.45 -> 3.65 ms over the seven expand blocks, bit-identical) — the per-wave loads with scalar offsets and an 8-deep ring hide their latency behind the
                    // 64-cycle fp32 MFMAs already, and the LDS reads compete with the plane traffic
                    static const int w1lds_env = sgx_getenv("SGX_IRB_W1LDS") ? atoi(sgx_getenv("SGX_IRB_W1LDS")) : 0;
                    if (ai >= 0 && w1lds_env) {
                        const int rows = (((ib.Cin >> 1) + 7) & ~7) * 2;
                        ib.w1rows = rows;
                        if (sgx_irb_lds_bytes(ib) > lds_max) ib.w1rows = 0;
                    }
                }
                {   // bf16x3 plan: every GEMM of the block needs its split weights (and the squeeze width must span the k16 steps the instantiation unrolls)
                    const int nqs = NQ == 0 ? 1 : (NQ == 2 ? 3 : (NT == 2 ? 1 : 2));
                    // k_irb3 is OPT-IN (SGX_DET_IRB3=1) until its operand streams hide their latency: measured slower than k_irb on every expand block (r4 trips: 0.99 against 0.75 ms on
                    // 112 -> 672 -> 112), equal on the SSD heads; the pointwise layers take the bf16x3 path by default
                    static const int irb3_env = sgx_getenv("SGX_DET_IRB3") ? atoi(sgx_getenv("SGX_DET_IRB3")) : 0;
                    const bool ok3 = irb3_env != 0 && h->gemm == 1 && c.wS && (ai < 0 || ops[ai].wS) && (di < 0 || (ops[di].wS && ops[ei].wS && (ops[di].outc + 15) / 16 == nqs));
                    ib.gemm = ok3 ? 1 : 0;
                    // mode 2: the fp32 block kernel with ONLY its expand GEMM as bf16x3 (SGX_DET_IRB_A3, tuning tap)
                    static const int a3_env = sgx_getenv("SGX_DET_IRB_A3") ? atoi(sgx_getenv("SGX_DET_IRB_A3")) : 0;
                    if (!ok3 && a3_env && h->gemm == 1 && ai >= 0 && ops[ai].wS) { ib.gemm = 2; ib.w1S = ops[ai].wS; }
                    if (ib.gemm == 2 && a3_env >= 2 && (a3_env == 2 || bq.outc == a3_env)) {      // experiment: pre-split input (SGX_DET_IRB_A3=2: every block; = Cexp: that block only)
                        ib.ldS = ((bq.H * bq.W + 31) / 32) * 32;
                        unsigned *sh = nullptr; if (h->alloc(&sh, (size_t)B * ((ops[ai].inc + 15) / 16) * 6 * ib.ldS * 4)) FAIL(SGX_ERR_NOMEM);      // 16 bytes per entry
                        ib.inS = sh;
                    } else if (ib.gemm == 2 && a3_env > 2 && bq.outc != a3_env) ib.gemm = 0;
                    if (ok3) { ib.w2S = c.wS; ib.w1S = ai >= 0 ? ops[ai].wS : nullptr; ib.wq1S = di >= 0 ? ops[di].wS : nullptr; ib.wq2S = di >= 0 ? ops[ei].wS : nullptr; }
                }
                {   // depthwise taps + bias, one padded row per channel
                    const int kk = bq.k * bq.k, rows = ((bq.outc + 31) / 32) * 32;
                    std::vector<float> wh((size_t)bq.outc * kk), bh(bq.outc), wp((size_t)rows * kkp, 0.f);
                    if (hipMemcpy(wh.data(), bq.wt, wh.size() * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(bh.data(), bq.bias, bh.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                    for (int m = 0; m < bq.outc; m++) { for (int t = 0; t < kk; t++) wp[(size_t)m * kkp + t] = wh[(size_t)m * kk + t]; wp[(size_t)m * kkp + kk] = bh[m]; }
                    float *dwp = nullptr; if (h->alloc(&dwp, wp.size())) FAIL(SGX_ERR_NOMEM);
                    if (hipMemcpy(dwp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                    ib.wdp = dwp;
                }
                Op f; f.kind = OP_IRB; f.in0 = ai >= 0 ? ops[ai].in0 : bq.in0; f.out = out_blob; f.irb = ib; f.irb_res_blob = res_blob;
                f.name = (ai >= 0 ? ops[ai].name + "+" : std::string()) + bq.name + "+" + c.name + (di >= 0 ? "+" + ops[di].name + "+" + ops[ei].name : std::string());
                f.inc = ib.Cin; f.outc = ib.Cout; f.H = bq.H; f.W = bq.W; f.Ho = bq.Ho; f.Wo = bq.Wo; f.k = bq.k; f.stride = bq.stride; f.hwc = c.hwc; f.hwc_off = c.hwc_off;
                // the block takes the place of its LAST convolution in the plan (every operand is older)
                const int last = di >= 0 ? ei : ci;
                if (ai >= 0) ops[ai].dead = true;
                bq.dead = true; if (last != ci) ops[ci].dead = true; if (di >= 0 && di != last) ops[di].dead = true;
                ops[last] = f;
            }
        // ---- squeeze-excite tails that are still two pointwise launches (the 38 x 38 blocks: 40 -> 10 -> 40 channels on 1 444 pixels) as one k_se_gate each
        {
            static const int seg_env = sgx_getenv("SGX_DET_SE_GATE") ? atoi(sgx_getenv("SGX_DET_SE_GATE")) : 1;
            for (int di = 0; seg_env && irb_mode == 1 && g_det_fuse && di < nops; di++) {
                Op &d = ops[di];
                if (d.dead || d.kind != OP_PW || d.hwc || d.in0 < 0) continue;
                const EpiClass cd = classify(d.epi);
                if (cd.mode != SGX_EMODE_ACT) continue;
                readers_all(d.out, R); if (R.size() != 1) continue;
                const int ei = R[0]; Op &e = ops[ei];
                if (e.kind != OP_PW || e.hwc || e.in0 != d.out || ei < di) continue;
                const EpiClass ce = classify(e.epi);
                if ((ce.mode != SGX_EMODE_GATE && ce.mode != SGX_EMODE_GATE_ADD) || ce.t0 != d.in0 || e.outc != d.inc || e.inc != d.outc) continue;
                if (!sgx_se_gate_supported(d.inc, d.outc) || d.out == h->loc_blob || d.out == h->conf_blob) continue;
                SgxSeGate sg; memset(&sg, 0, sizeof sg);
                sg.Cout = d.inc; sg.Cq = d.outc; sg.HW = d.H * d.W;
                sg.se.bq1 = d.bias; sg.se.bq2 = e.bias; sg.se.qlo = cd.lo; sg.se.qhi = cd.hi; sg.se.gc1 = ce.c1; sg.se.glo = ce.lo; sg.se.ghi = ce.hi; sg.se.gc2 = ce.c2; sg.wq1 = d.wt; sg.wq2 = e.wt;
                {
                    const int Cq = d.outc, Co = d.inc;
                    std::vector<float> q1((size_t)Cq * Co), q2((size_t)Co * Cq), p1((size_t)Cq * Co), p2((size_t)Co * Cq);
                    if (hipMemcpy(q1.data(), d.wt, q1.size() * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(q2.data(), e.wt, q2.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                    for (int j = 0; j < Cq; j++) for (int k = 0; k < Co; k++) p1[((size_t)(j >> 1) * Co + k) * 2 + (j & 1)] = q1[(size_t)j * Co + k];
                    for (int co = 0; co < Co; co++) for (int j = 0; j < Cq; j++) p2[((size_t)j * (Co / 2) + (co >> 1)) * 2 + (co & 1)] = q2[(size_t)co * Cq + j];
                    float *dp1 = nullptr, *dp2 = nullptr;
                    if (h->alloc(&dp1, p1.size()) || h->alloc(&dp2, p2.size())) FAIL(SGX_ERR_NOMEM);
                    if (hipMemcpy(dp1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dp2, p2.data(), p2.size() * 4, hipMemcpyHostToDevice) != hipSuccess) FAIL(SGX_ERR_DEVICE);
                    sg.se.wq1p = dp1; sg.se.wq2p = dp2;
                }
                Op f; f.kind = OP_SE_GATE; f.in0 = d.in0; f.out = e.out; f.sg = sg; f.sg_res_blob = ce.mode == SGX_EMODE_GATE_ADD ? ce.t1 : -1;
                f.name = d.name + "+" + e.name; f.inc = d.inc; f.outc = e.outc; f.H = d.H; f.W = d.W; f.Ho = d.H; f.Wo = d.W;
                d.dead = true; ops[ei] = f;
            }
        }
        }
        // ---- the two SSD heads of a feature map (loc and conf: depthwise 3x3 + ReLU -> pointwise, HWC store) read the same planes: one kernel stages them once, reads the
        // depthwise taps once and runs both heads' weights over them (k_irb with a second accumulator set).  SGX_DET_IRB_DUAL=0 keeps them apart.
        {
            static const int dual_env = sgx_getenv("SGX_DET_IRB_DUAL") ? atoi(sgx_getenv("SGX_DET_IRB_DUAL")) : 1;
            for (int i = 0; dual_env && i < nops; i++) {
                Op &a = ops[i];
                if (a.dead || a.kind != OP_IRB || a.irb.has_expand || !a.irb.hwc || a.irb.Cq || a.irb.Cout2) continue;
                for (int j = 0; j < nops; j++) {
                    Op &b = ops[j];
                    if (j == i || b.dead || b.kind != OP_IRB || b.irb.has_expand || !b.irb.hwc || b.irb.Cq || b.irb.Cout2 || b.in0 != a.in0) continue;
                    if (b.irb.K != a.irb.K || b.irb.S != a.irb.S || b.irb.Cexp != a.irb.Cexp || b.irb.H != a.irb.H || b.irb.W != a.irb.W || b.irb.act2 != a.irb.act2 ||
                        b.irb.a2lo != a.irb.a2lo || b.irb.a2hi != a.irb.a2hi || b.irb.G != a.irb.G || b.irb.nbands != a.irb.nbands) continue;
                    if (b.irb.Cout > 32 || b.irb.Cout > a.irb.Cout || b.irb.gemm != a.irb.gemm) continue;                 // the narrow (loc) head rides along as the second accumulator set
                    if (!sgx_irb_supported(a.irb.K, a.irb.S, (a.irb.Cout + 31) / 32, 0, false, a.irb.act2 == SGX_EMODE_HSWISH, 1)) continue;
                    SgxIrb m = a.irb;
                    m.Cout2 = b.irb.Cout; m.hwc_off2 = b.irb.hwc_off; m.ld2b = b.irb.ld2; m.wdp2 = b.irb.wdp; m.w2Tb = b.irb.w2T; m.b2b = b.irb.b2; m.w2Sb = b.irb.w2S;
                    m.wd_b = b.irb.wd; m.bd_b = b.irb.bd; m.w2_b = b.irb.w2;
                    if (sgx_irb_lds_bytes(m) > 160 * 1024) { m.nbuf = 1; if (sgx_irb_lds_bytes(m) > 160 * 1024) continue; }
                    Op f = a; f.irb = m; f.irb_out2_blob = b.out; f.name = a.name + "|" + b.name;
                    const int last = std::max(i, j);
                    ops[i].dead = true; ops[j].dead = true; ops[last] = f; ops[last].dead = false;
                    break;
                }
            }
        }
        std::vector<Op> live; for (const Op &o : ops) if (!o.dead) live.push_back(o);
        ops.swap(live);
    }
    for (const Op &o : h->ops) {
        Blob &ob = h->blobs[o.out]; if (!ob.d && ob.n) { if (h->alloc(&ob.d, ob.n * B)) FAIL(SGX_ERR_NOMEM); }
        if (o.irb_out2_blob >= 0) { Blob &o2 = h->blobs[o.irb_out2_blob]; if (!o2.d && o2.n) { if (h->alloc(&o2.d, o2.n * B)) FAIL(SGX_ERR_NOMEM); } }
    }
    {   // pre-processing fused into the stem (k_stem_pre) when the stem is the only reader of the network input and takes the k_conv_stem2 path (see run_op)
        static const int prefuse_env = sgx_getenv("SGX_DET_PREFUSE") ? atoi(sgx_getenv("SGX_DET_PREFUSE")) : 1;
        const int in_id = h->blob_id.at("input");
        int readers = 0; for (const Op &o : h->ops) readers += (o.in0 == in_id) + (o.in1 == in_id);
        if (prefuse_env && g_det_fuse && !h->legacy && !h->ops.empty() && readers == 1 && stem2_ok(h, h->ops[0]) && h->ops[0].in0 == in_id) {
            Blob &ib = h->blobs[in_id];
            auto it = std::find(h->dev.begin(), h->dev.end(), (void *)ib.d);
            if (it != h->dev.end()) { (void)hipFree(ib.d); h->dev.erase(it); ib.d = nullptr; h->pre_fused = 1; }
        }
    }
#undef FAIL
    h->num_priors = (int)prior_boxes.size() / 4;
    h->priors = prior_boxes; h->priors.insert(h->priors.end(), prior_vars.begin(), prior_vars.end());
    if ((size_t)h->num_priors * 4 != h->blobs[h->loc_blob].n || (size_t)h->num_priors * h->num_class != h->blobs[h->conf_blob].n) { delete h; return SGX_ERR_INVALID; }
    if (h->num_priors > SGX_DO_SORT || h->nms_top_k > SGX_DO_TOPK || (h->num_class - 1) * h->nms_top_k > SGX_DO_MERGE || h->num_class - 1 > 63 || h->num_class < 2 ||
        h->keep_top_k > SGX_DET_MAX) { delete h; return SGX_ERR_UNSUPPORTED; }
    if (h->alloc(&h->d_priors, (size_t)h->num_priors * 4) || h->alloc(&h->d_cls_rows, (size_t)B * (h->num_class - 1) * SGX_DO_TOPK * 6) ||
        h->alloc(&h->d_cls_count, (size_t)B * (h->num_class - 1)) || h->alloc(&h->d_results, (size_t)B)) { delete h; return SGX_ERR_NOMEM; }
    if (hipMemcpy(h->d_priors, prior_boxes.data(), sizeof(float) * 4 * h->num_priors, hipMemcpyHostToDevice) != hipSuccess) { delete h; return SGX_ERR_DEVICE; }
    {   // A/B switch of the XCD-aware work order (sgx_xcd_order); on by default
        const int xo = sgx_getenv("SGX_DET_XCD") ? atoi(sgx_getenv("SGX_DET_XCD")) : 1;
#ifndef SGX_EMU
        if (hipMemcpyToSymbol(HIP_SYMBOL(sgx_det_xcd_order), &xo, sizeof(int)) != hipSuccess) { delete h; return SGX_ERR_DEVICE; }
#else
        sgx_det_xcd_order = xo;
#endif
    }
    if (hipMemset(h->d_results, 0, sizeof(sgx_det_result) * (size_t)B) != hipSuccess) { delete h; return SGX_ERR_DEVICE; }      // entries past the counts are never written: keep them defined
    std::vector<SgxDetTab> xt, yt; build_tab(width, T, xt); build_tab(height, T, yt);
    if (h->alloc(&h->d_xt, T) || h->alloc(&h->d_yt, T) || h->alloc(&h->d_img, (size_t)B * height * ((3 * width + 3) & ~3) + 4)) { delete h; return SGX_ERR_NOMEM; }
    if (hipMemcpy(h->d_xt, xt.data(), sizeof(SgxDetTab) * T, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(h->d_yt, yt.data(), sizeof(SgxDetTab) * T, hipMemcpyHostToDevice) != hipSuccess) { delete h; return SGX_ERR_DEVICE; }
    *out = h;
    return SGX_OK;
}

static SgxEpi make_epi(const sgx_det *h, const Op &op, size_t tpitch)
{
    SgxEpi e; memset(&e, 0, sizeof e);
    e.tpitch = tpitch;
    if (op.act == SGX_ACT_RELU) { e.s[e.n].op = SGX_EOP_RELU; e.n++; }
    else if (op.act == SGX_ACT_CLIP) { e.s[e.n].op = SGX_EOP_CLIP; e.s[e.n].a = op.lo; e.s[e.n].b = op.hi; e.n++; }
    for (const EpiStep &st : op.epi) {
        SgxEpiStep &d = e.s[e.n++];
        d.op = st.op; d.src = st.src; d.a = st.a; d.b = st.b; d.t = st.tensor >= 0 ? h->blobs[st.tensor].d : nullptr;
    }
    // recognise the recurring programs (straight-line code in the kernels); anything else runs through the generic interpreter
    auto is = [&](int i, int opc, int src) { return i < e.n && e.s[i].op == opc && (opc == SGX_EOP_CLIP || opc == SGX_EOP_RELU || e.s[i].src == src); };
    const float inf = INFINITY;
    e.mode = SGX_EMODE_GENERIC;
    if (e.n == 0) e.mode = SGX_EMODE_NONE;
    else if (e.n == 1 && is(0, SGX_EOP_RELU, 0)) { e.mode = SGX_EMODE_ACT; e.lo = 0.f; e.hi = inf; }
    else if (e.n == 1 && is(0, SGX_EOP_CLIP, 0)) { e.mode = SGX_EMODE_ACT; e.lo = e.s[0].a; e.hi = e.s[0].b; }
    else if (e.n == 1 && is(0, SGX_EOP_ADD, SGX_ESRC_TENSOR)) { e.mode = SGX_EMODE_ADD_T; e.t1 = e.s[0].t; }
    else if (e.n == 4 && is(0, SGX_EOP_ADD, SGX_ESRC_CONST) && is(1, SGX_EOP_CLIP, 0) && is(2, SGX_EOP_MUL, SGX_ESRC_ROOT) && is(3, SGX_EOP_DIV, SGX_ESRC_CONST)) {
        e.mode = SGX_EMODE_HSWISH; e.c1 = e.s[0].a; e.lo = e.s[1].a; e.hi = e.s[1].b; e.c2 = e.s[3].a;
    } else if ((e.n == 4 || e.n == 5) && is(0, SGX_EOP_ADD, SGX_ESRC_CONST) && is(1, SGX_EOP_CLIP, 0) && is(2, SGX_EOP_DIV, SGX_ESRC_CONST) && is(3, SGX_EOP_MUL, SGX_ESRC_TENSOR) &&
               (e.n == 4 || is(4, SGX_EOP_ADD, SGX_ESRC_TENSOR))) {
        e.mode = e.n == 4 ? SGX_EMODE_GATE : SGX_EMODE_GATE_ADD; e.c1 = e.s[0].a; e.lo = e.s[1].a; e.hi = e.s[1].b; e.c2 = e.s[2].a; e.t0 = e.s[3].t; if (e.n == 5) e.t1 = e.s[4].t;
    }
    return e;
}

#ifdef SGX_EMU
// Software model of k_conv_pw3 (tests only): the same operands the device kernel consumes — the host-split weights READ BACK from their matrix-core layout, the activations split into
// three bf16 terms (round to nearest even, exact residuals) — the six leading cross terms per k16 step in the kernel's order, each term's sixteen exact products summed wide and
// rounded into the fp32 accumulator once (a model of one MFMA: the hardware's internal order is not specified; any fp32-grade order meets the tests' drift criterion), rows past
// the last input channel clamped (they meet zero weights).  Epilogue as the device.
static void sgx_pw3_emu(int inc, int outc, int N, int batch, const float *in, size_t in_pitch, const unsigned short *wS, int ldw, const float *bias, float *out, size_t out_pitch,
                        const SgxEpi &epi, int hwc, int hwc_off)
{
    const int nks = (inc + 15) / 16;
    auto split = [](float x, float *t) {
        const unsigned short h0 = sgx_bf16_rne(x); t[0] = sgx_bf16_to_f32(h0); const float r1 = x - t[0];
        const unsigned short h1 = sgx_bf16_rne(r1); t[1] = sgx_bf16_to_f32(h1); t[2] = sgx_bf16_to_f32(sgx_bf16_rne(r1 - t[1]));
    };
    std::vector<float> xs((size_t)nks * 16 * 3);
    for (int b = 0; b < batch; b++) for (int n = 0; n < N; n++) {
        const float *X = in + (size_t)b * in_pitch;
        for (int k = 0; k < nks * 16; k++) split(X[(size_t)std::min(k, inc - 1) * N + n], &xs[(size_t)k * 3]);
        for (int o = 0; o < outc; o++) {
            float acc = bias[o];
            for (int s = 0; s < nks; s++) {
                static const int TA[6] = { 0, 1, 2, 0, 1, 0 }, TB[6] = { 2, 1, 0, 1, 0, 0 };
                for (int q = 0; q < 6; q++) {
                    double sum = 0;
                    for (int kk = 0; kk < 16; kk++) {
                        const int hf = kk >> 3, j = kk & 7;
                        const float a = sgx_bf16_to_f32(wS[((((size_t)s * 3 + TA[q]) * 2 + hf) * ldw + o) * 8 + j]);
                        sum += (double)a * (double)xs[(size_t)(16 * s + kk) * 3 + TB[q]];
                    }
                    acc = (float)((double)acc + sum);
                }
            }
            const float v = sgx_epi(epi, acc, (size_t)b * epi.tpitch + (size_t)o * N + n);
            float *Y = out + (size_t)b * out_pitch;
            if (hwc) Y[(size_t)hwc_off + (size_t)n * outc + o] = v; else Y[(size_t)o * N + n] = v;
        }
    }
}
#endif

static void run_op(sgx_det *h, const Op &op, int batch, sgx_stream_t st)
{
    const Blob &A = h->blobs[op.in0]; const Blob &O = h->blobs[op.out];
    switch (op.kind) {
    case OP_PW: {
        const int N = op.H * op.W;
        const size_t big = (size_t)batch * std::max(std::max(A.n, O.n), (size_t)op.outc * N) * 4;      // k_conv_pw2 uses 32-bit byte offsets per lane
        if (h->legacy || !op.wtT || (op.inc & 1) || big >= 0xFFFFFFFFull) {
            SGX_LAUNCH(k_conv_pw, dim3((N + 63) / 64, (op.outc + 63) / 64, batch), dim3(256), st, op.inc, op.outc, N, A.d, A.n, op.wt, op.bias, O.d, O.n,
                       make_epi(h, op, (size_t)op.outc * N), op.hwc, op.hwc_off);
            break;
        }
        static const int pw3_mink = sgx_getenv("SGX_PW3_MINK") ? atoi(sgx_getenv("SGX_PW3_MINK")) : 64;      // measured: with fewer than four k16 steps the exact-fp32 kernel's shorter prologue wins (c40 -> 120 / 160: 0.14 against 0.17 ms)
#ifdef SGX_EMU
        if (h->gemm == 1 && op.wS && op.inc >= pw3_mink) {
            sgx_pw3_emu(op.inc, op.outc, N, batch, A.d, A.n, (const unsigned short *)op.wS, op.ldw, op.bias, O.d, O.n, make_epi(h, op, (size_t)op.outc * N), op.hwc, op.hwc_off);
            break;
        }
#else
        if (h->gemm == 1 && op.wS && op.inc >= pw3_mink) {
            // bf16x3 (k_conv_pw3): same decomposition; the accumulators + the split operands cap the wave tile at four 32 x 32 sub-tiles
            const int sub = (op.outc + 31) / 32, total = batch * N;
            static const int cand3[8][2] = { {2, 2}, {4, 1}, {3, 1}, {5, 1}, {1, 4}, {2, 1}, {1, 2}, {1, 1} };
            static const int force3 = sgx_getenv("SGX_PW3_FORCE") ? atoi(sgx_getenv("SGX_PW3_FORCE")) : 0;      // tuning tap: OCB * 10 + PXB
            int ocb = 1, pxb = 1; long best_score = -1;
            for (int c = 0; c < 8; c++) {
                const int cb = cand3[c][0], cp = cand3[c][1];
                const long nwg = (long)((total + 128 * cp - 1) / (128 * cp)) * ((sub + cb - 1) / cb);
                const int padded = ((sub + cb - 1) / cb) * cb;
                const long fill = std::min(nwg, 512L);
                const long score = fill * 1000000L + (long)(1000 - (padded - sub) * 100) * 100L + cb * 20 + cp;        // among equals: more oc tiles per operand split (the split is the vector work of this kernel)
                if (score > best_score) { best_score = score; ocb = cb; pxb = cp; }
            }
            if (force3) { ocb = force3 / 10; pxb = force3 % 10; }
            const int nxt = (total + 128 * pxb - 1) / (128 * pxb), noc = (sub + ocb - 1) / ocb;
            const int grid = ((nxt + 7) / 8) * 8 * noc;
            const SgxEpi e = make_epi(h, op, (size_t)op.outc * N);
#define SGX_PW3(OCB_, PXB_) do { auto kfn = k_conv_pw3<OCB_, PXB_>; SGX_LAUNCH(kfn, dim3(grid), dim3(256), st, op.inc, op.outc, N, total, A.d, A.n, (const sgx_u32x4 *)op.wS, op.bias, \
                                                                               O.d, O.n, e, op.hwc, op.hwc_off, nxt, noc, op.ldw, 1); } while (0)
            switch (ocb * 10 + pxb) {
            case 22: SGX_PW3(2, 2); break; case 41: SGX_PW3(4, 1); break; case 31: SGX_PW3(3, 1); break; case 51: SGX_PW3(5, 1); break; case 14: SGX_PW3(1, 4); break;
            case 21: SGX_PW3(2, 1); break; case 12: SGX_PW3(1, 2); break; default: SGX_PW3(1, 1); break;
            }
#undef SGX_PW3
            break;
        }
#endif
        // Tile choice.  oc block = OCB sub-tiles of 32 channels, wave tile = PXB sub-tiles of 32 pixels, workgroup = 4 waves along pixels.
        // Largest tile (most operand reuse) whose grid still gives every CU >= 2 workgroups, among those with the least oc padding;
        // if no candidate fills the chip, the one with the most workgroups.
        const int sub = (op.outc + 31) / 32, total = batch * N;
        static const int cand[7][2] = { {4, 2}, {3, 2}, {2, 2}, {1, 4}, {2, 1}, {1, 2}, {1, 1} };
        int ocb = 1, pxb = 1; long best_score = -1;
        for (int c = 0; c < 7; c++) {
            const int cb = cand[c][0], cp = cand[c][1];
            const long nwg = (long)((total + 128 * cp - 1) / (128 * cp)) * ((sub + cb - 1) / cb);
            const int padded = ((sub + cb - 1) / cb) * cb;
            // score: filling the chip first (capped), then little padding, then tile size
            const long fill = std::min(nwg, 512L);
            const long score = fill * 1000000L + (long)(1000 - (padded - sub) * 100) * 100L + cb * cp;
            if (score > best_score) { best_score = score; ocb = cb; pxb = cp; }
        }
        const int nxt = (total + 128 * pxb - 1) / (128 * pxb), noc = (sub + ocb - 1) / ocb;
        const int grid = ((nxt + 7) / 8) * 8 * noc;
        const SgxEpi e = make_epi(h, op, (size_t)op.outc * N);
        static const int pw2_direct = sgx_getenv("SGX_PW2_DIRECT") ? atoi(sgx_getenv("SGX_PW2_DIRECT")) : 1;
#define SGX_PW2(OCB_, PXB_) do { auto kfn = k_conv_pw2<OCB_, PXB_>; SGX_LAUNCH(kfn, dim3(grid), dim3(256), st, op.inc, op.outc, N, total, A.d, A.n, op.wtT, op.bias, \
                                                                               O.d, O.n, e, op.hwc, op.hwc_off, nxt, noc, op.ldw, pw2_direct); } while (0)
        switch (ocb * 10 + pxb) {
        case 42: SGX_PW2(4, 2); break; case 32: SGX_PW2(3, 2); break; case 22: SGX_PW2(2, 2); break; case 14: SGX_PW2(1, 4); break;
        case 21: SGX_PW2(2, 1); break; case 12: SGX_PW2(1, 2); break; default: SGX_PW2(1, 1); break;
        }
#undef SGX_PW2
        break; }
    case OP_FUSED_BLOCK: {
        SgxFusedBlk fb = op.fb;
        fb.in = A.d; fb.in_pitch = A.n; fb.out = O.d; fb.out_pitch = O.n;
        fb.res = op.fb_res_blob >= 0 ? h->blobs[op.fb_res_blob].d : nullptr; fb.res_pitch = op.fb_res_blob >= 0 ? h->blobs[op.fb_res_blob].n : 0;
        if (fb.hrb) {
            SgxHrb q; memset(&q, 0, sizeof q);
            q.Cin = fb.Cin; q.Cmid = fb.Cmid; q.Cout = fb.Cout; q.Cq = fb.Cq; q.K = fb.K; q.S = fb.stride; q.pad = fb.pad; q.H = fb.H; q.W = fb.W; q.Ho = fb.Ho; q.Wo = fb.Wo;
            q.TOH = fb.TOH; q.TOW = fb.TOW; q.occ = fb.hrb_occ; q.tiles_x = fb.tiles_x; q.tiles_y = fb.tiles_y; q.lo1 = fb.lo1; q.hi1 = fb.hi1; q.lo2 = fb.lo2; q.hi2 = fb.hi2;
            q.in = fb.in; q.in_pitch = fb.in_pitch; q.out = fb.out; q.out_pitch = fb.out_pitch; q.res = fb.res; q.res_pitch = fb.res_pitch;
            q.w1S = (const sgx_q4 *)fb.w1S; q.ld1 = fb.ld1S; q.b1 = fb.b1; q.wd2 = fb.wd2; q.bd = fb.bd; q.w2S = (const sgx_q4 *)fb.w2S; q.ld2 = fb.ld2S; q.b2 = fb.b2;
            q.wq1S = (const sgx_q4 *)fb.wq1S; q.wq2S = (const sgx_q4 *)fb.wq2S; q.ldq1 = fb.ldq1S; q.ldq2 = fb.ldq2S; q.bq1 = fb.bq1; q.bq2 = fb.bq2;
            q.qlo = fb.qlo; q.qhi = fb.qhi; q.gc1 = fb.gc1; q.glo = fb.glo; q.ghi = fb.ghi; q.gc2 = fb.gc2;
            (void)sgx_hrb_launch(q, batch, st); break;
        }
        if (fb.v2) { (void)sgx_fb2_launch(fb, batch, st); break; }                      // the variant was validated when the plan was built
#ifdef SGX_DEBUG_TAPS
        SGX_LAUNCH_DYN(k_fused_block, dim3((unsigned)(fb.tiles_x * fb.tiles_y * batch)), dim3(256), sgx_fb_lds_floats(fb) * 4, st, fb);
#endif
        break; }
    case OP_SE_GATE: {
        SgxSeGate sg = op.sg;
        sg.y = A.d; sg.y_pitch = A.n; sg.out = O.d; sg.out_pitch = O.n;
        sg.res = op.sg_res_blob >= 0 ? h->blobs[op.sg_res_blob].d : nullptr; sg.res_pitch = op.sg_res_blob >= 0 ? h->blobs[op.sg_res_blob].n : 0;
        (void)sgx_se_gate_launch(sg, batch, st);
        break; }
    case OP_IRB: {
        SgxIrb ib = op.irb;
        ib.in = A.d; ib.in_pitch = A.n; ib.out = O.d; ib.out_pitch = O.n;
        ib.res = op.irb_res_blob >= 0 ? h->blobs[op.irb_res_blob].d : nullptr; ib.res_pitch = op.irb_res_blob >= 0 ? h->blobs[op.irb_res_blob].n : 0;
        if (op.irb_out2_blob >= 0) { ib.out2 = h->blobs[op.irb_out2_blob].d; ib.out2_pitch = h->blobs[op.irb_out2_blob].n; }
        (void)sgx_irb_launch(ib, batch, st);                                             // the instantiation was validated when the plan was built
        break; }
    case OP_KXK: {
        const SgxEpi e = make_epi(h, op, (size_t)op.outc * op.Ho * op.Wo);
        static const int budget_env = sgx_getenv("SGX_DW_BUDGET") ? atoi(sgx_getenv("SGX_DW_BUDGET")) : 8192;
        const int budget = budget_env;                                  // floats of LDS per workgroup for the staged input
        const int Wp = (op.Wo - 1) * op.stride + op.k;
        auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
        const int nbx4 = (op.Wo + 3) / 4, pitch4 = ((nbx4 - 1) * 4 * op.stride + 3 * op.stride + op.k + 3) & ~3;
        static const int dw2_on = sgx_getenv("SGX_DW2") ? atoi(sgx_getenv("SGX_DW2")) : 1;
#ifdef SGX_DEBUG_TAPS
        static const int dw3_on = sgx_getenv("SGX_DW3") ? atoi(sgx_getenv("SGX_DW3")) : 0;      // round 6 experiment: k_conv_dw3 (channel pairs, packed FMAs) measured SLOWER than k_conv_dw2 (0.26 vs 0.18 ms on the 38 x 38 planes): tap only
        if (dw3_on && !h->legacy && op.depthwise && op.wtP) {
            SgxDw3 d; int px = 1; size_t lds3 = 0;
            if (sgx_dw3_plan(op.outc, op.H, op.W, op.Ho, op.Wo, op.k, op.stride, op.pad, e.mode, &d, &px, &lds3)) {
                d.in = A.d; d.in_pitch = A.n; d.wd2 = op.wtP; d.bias = op.bias; d.out = O.d; d.out_pitch = O.n; d.epi = e;
                sgx_dw3_launch(d, op.k, op.stride, px, lds3, batch, st);
                break;
            }
        }
#endif
        if (dw2_on && !h->legacy && op.depthwise && (op.k == 3 || op.k == 5) && (op.stride == 1 || op.stride == 2) && pitch4 * op.k <= budget) {
            // k_conv_dw2: P planes x a band of RB output rows per workgroup, LDS tile [P][(RB - 1) s + k][pitch4]
            const int nplanes = batch * op.outc, rin_full = (op.Ho - 1) * op.stride + op.k, KW = (op.k * op.k + 1 + 3) & ~3;
            int P = 1, RB = op.Ho;
            if (rin_full * pitch4 <= budget) P = std::max(1, std::min(std::min(budget / (rin_full * pitch4), 16), nplanes / 2048));
            else {
                const int rbmax = std::max(1, (budget / pitch4 - op.k) / op.stride + 1);
                int nb = (op.Ho + rbmax - 1) / rbmax;
                nb = std::max(nb, std::min((2048 + nplanes - 1) / nplanes, std::max(1, op.Ho / 4)));
                RB = (op.Ho + nb - 1) / nb;
            }
            const int nbands = (op.Ho + RB - 1) / RB, ngroups = (nplanes + P - 1) / P;
            const size_t lds = ((size_t)P * ((RB - 1) * op.stride + op.k) * pitch4 + (size_t)P * KW) * 4;
#define SGX_DW2(K_, S_) do { auto kfn = k_conv_dw2<K_, S_>; SGX_LAUNCH_DYN(kfn, dim3(ngroups * nbands), dim3(256), lds, st, op.outc, op.H, op.W, op.Ho, op.Wo, op.pad, P, RB, nbands, nplanes, pitch4, \
                                                                          magic(std::min(RB, op.Ho) * nbx4), magic(nbx4), A.d, op.wt, op.bias, O.d, e); } while (0)
            if (op.k == 3 && op.stride == 1) SGX_DW2(3, 1); else if (op.k == 3) SGX_DW2(3, 2); else if (op.stride == 1) SGX_DW2(5, 1); else SGX_DW2(5, 2);
#undef SGX_DW2
        } else if (!h->legacy && op.depthwise && (op.k == 3 || op.k == 5) && Wp * op.k <= budget) {
            const int nplanes = batch * op.outc, rin_full = (op.Ho - 1) * op.stride + op.k;
            int P = 1, RB = op.Ho;
            if (rin_full * Wp <= budget) P = std::max(1, std::min(std::min(budget / (rin_full * Wp), 16), nplanes / 2048));
            else {
                const int rbmax = std::max(1, (budget / Wp - op.k) / op.stride + 1);
                int nb = (op.Ho + rbmax - 1) / rbmax;
                nb = std::max(nb, std::min((2048 + nplanes - 1) / nplanes, std::max(1, op.Ho / 4)));
                RB = (op.Ho + nb - 1) / nb;
            }
            const int nbands = (op.Ho + RB - 1) / RB, ngroups = (nplanes + P - 1) / P;
            const size_t lds = ((size_t)P * ((RB - 1) * op.stride + op.k) * Wp + (size_t)P * (op.k * op.k + 1)) * 4;
            if (op.k == 3) { auto kfn = k_conv_dw<3>; SGX_LAUNCH_DYN(kfn, dim3(ngroups * nbands), dim3(256), lds, st, op.outc, op.H, op.W, op.Ho, op.Wo, op.stride, op.pad, P, RB, nbands, nplanes,
                                                                     magic(((std::min(RB, op.Ho) - 1) * op.stride + op.k) * Wp), magic(Wp), magic(op.Wo), A.d, op.wt, op.bias, O.d, e); }
            else { auto kfn = k_conv_dw<5>; SGX_LAUNCH_DYN(kfn, dim3(ngroups * nbands), dim3(256), lds, st, op.outc, op.H, op.W, op.Ho, op.Wo, op.stride, op.pad, P, RB, nbands, nplanes,
                                                           magic(((std::min(RB, op.Ho) - 1) * op.stride + op.k) * Wp), magic(Wp), magic(op.Wo), A.d, op.wt, op.bias, O.d, e); }
        } else if (stem2_ok(h, op)) {
            // k_conv_stem2: bands of RB output rows; RB*ceil(Wo/4) tasks for 256 threads
            const Stem2Geom g = stem2_geom(op);
            auto kfn = k_conv_stem2<3>;
            SGX_LAUNCH_DYN(kfn, dim3(g.nbands, batch), dim3(256), g.lds, st, op.outc, op.H, op.W, op.Ho, op.Wo, op.pad, g.RB, g.pitch4, magic(g.nbx4), A.d, A.n, op.wtT, op.bias, O.d, O.n, e);
        } else if (!h->legacy && !op.depthwise && op.wtT && op.outc <= 16 && op.inc * Wp * op.k <= budget) {
            const int RB = std::min(op.Ho, std::max(1, (budget / (op.inc * Wp) - op.k) / op.stride + 1)), nbands = (op.Ho + RB - 1) / RB;
            const size_t lds = (size_t)op.inc * ((RB - 1) * op.stride + op.k) * Wp * 4;
            SGX_LAUNCH_DYN(k_conv_stem, dim3(nbands, batch), dim3(256), lds, st, op.inc, op.outc, op.H, op.W, op.Ho, op.Wo, op.k, op.stride, op.pad, RB, magic(((RB - 1) * op.stride + op.k) * Wp), magic(Wp), magic(op.Wo),
                           A.d, A.n, op.wtT, op.bias, O.d, O.n, e);
        } else
            SGX_LAUNCH(k_conv_kxk, dim3((op.Ho * op.Wo + 255) / 256, op.outc, batch), dim3(256), st, op.inc, op.outc, op.H, op.W, op.Ho, op.Wo, op.k, op.stride, op.pad, op.depthwise,
                       A.d, A.n, op.wt, op.bias, O.d, O.n, e);
        break; }
    case OP_BINARY: {
        const Blob &Bb = h->blobs[op.in1];
        // per-image pitch equals blob size (dense), so the batch is one flat range
        const size_t n = A.n * batch; const int g = (int)std::min<size_t>((n + 255) / 256, 8192);
        SGX_LAUNCH(k_binary, dim3(g), dim3(256), st, n, op.bop, A.d, Bb.scalar ? A.d : Bb.d, Bb.scalar ? 1 : 0, Bb.sval, O.d);
        break; }
    case OP_UNARY: { const size_t n = A.n * batch; const int g = (int)std::min<size_t>((n + 255) / 256, 8192); SGX_LAUNCH(k_unary, dim3(g), dim3(256), st, n, op.act, op.lo, op.hi, A.d, O.d); break; }
    case OP_PERMUTE_INTO: SGX_LAUNCH(k_permute_hwc_into, dim3((op.C * op.rows + 255) / 256, batch), dim3(256), st, op.C, op.rows, A.d, A.n, O.d, O.n, op.off); break;
    case OP_COPY_INTO: SGX_LAUNCH(k_copy_into, dim3(((int)A.n + 255) / 256, batch), dim3(256), st, (int)A.n, A.d, A.n, O.d, O.n, op.off); break;
    case OP_SOFTMAX: SGX_LAUNCH(k_softmax_rows, dim3((op.rows + 255) / 256, batch), dim3(256), st, op.rows, op.C, A.d, A.n, O.d, O.n); break;
    }
}

static void run_preprocess(sgx_det *h, const uint8_t *d_img, int pitch, int batch, sgx_stream_t st)
{
    const int T = h->T;
    SGX_LAUNCH(k_det_preprocess, dim3(T, batch), dim3(256), st, batch, d_img, h->W, h->H, pitch, h->d_xt, h->d_yt, T, 123.675f, 116.28f, 103.53f,
               h->blobs[h->blob_id.at("input")].d);
}

static bool stem2_epi_ok(const sgx_det *h, const Op &op)
{
    const SgxEpi e = make_epi(h, op, 0);
    return e.mode == SGX_EMODE_NONE || e.mode == SGX_EMODE_ACT || e.mode == SGX_EMODE_HSWISH;
}

// first step of the plan: k_det_preprocess into the "input" blob, or (pre_fused) the stem convolution straight from the images.  Launched outside the captured
// graph: the image pointer changes from call to call.
static void run_first_step(sgx_det *h, const uint8_t *d_img, int pitch, int batch, sgx_stream_t st)
{
    if (!h->pre_fused) { run_preprocess(h, d_img, pitch, batch, st); return; }
    const Op &op = h->ops[0]; const Blob &O = h->blobs[op.out];
    const SgxEpi e = make_epi(h, op, (size_t)op.outc * op.Ho * op.Wo);
    const Stem2Geom g = stem2_geom(op);
    auto magic = [](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    SGX_LAUNCH_DYN(k_stem_pre, dim3(g.nbands * batch), dim3(256), g.lds + (size_t)(h->T + (g.RB - 1) * 2 + 3) * sizeof(SgxDetTab), st, d_img, h->H, pitch, h->d_xt, h->d_yt, h->T, 123.675f, 116.28f, 103.53f, g.nbands, batch,
                   op.outc, op.Ho, op.Wo, op.pad, g.RB, g.pitch4, magic(g.pitch4), magic(g.nbx4), op.wtT, op.bias, O.d, O.n, e);
}

#ifndef SGX_EMU
// The plan as a hipGraph WITH its parallel branches.  A one-stream capture turns the plan into a chain of nodes, so the SSD heads of a feature map (latency-bound: a few deep-k
// workgroup chains per image) hold up the backbone layers behind them although nothing there reads their output, and the extra-layer tail (5 x 5 ... 1 x 1 maps) runs alone on the
// chip.  Here the capture forks: dependencies between plan steps are derived from the blobs they read and write (read-after-write, write-after-read, write-after-write; the head
// kernels' HWC stores into the shared concat buffers are disjoint slices and do not order each other), steps are dealt to `lanes` capture streams — a step continues the lane of
// the producer whose longest remaining path runs through it (the backbone stays on lane 0), otherwise it takes a free side lane — and every cross-lane dependency becomes an
// event edge.  All lanes rejoin lane 0 before the capture ends.  Same kernels, same arguments, same results; only the order constraints the hardware sees are fewer.
static int capture_forked(sgx_det *h, size_t first, int batch, sgx_stream_t st, int lanes, hipGraph_t *graph)
{
    const int n = (int)h->ops.size();
    auto root = [&](int b) { while (b >= 0 && h->blobs[b].alias >= 0) b = h->blobs[b].alias; return b; };
    std::vector<std::vector<int>> rd(n), wr(n); std::vector<char> partial(n, 0);
    for (int i = (int)first; i < n; i++) {
        const Op &o = h->ops[i];
        for (int b : { o.in0, o.in1, o.fb_res_blob, o.sg_res_blob, o.irb_res_blob }) if (b >= 0 && !h->blobs[b].scalar) rd[i].push_back(root(b));
        for (const EpiStep &e : o.epi) if (e.tensor >= 0) rd[i].push_back(root(e.tensor));
        wr[i].push_back(root(o.out)); if (o.irb_out2_blob >= 0) wr[i].push_back(root(o.irb_out2_blob));
        partial[i] = (o.hwc || o.kind == OP_PERMUTE_INTO || o.kind == OP_COPY_INTO || (o.kind == OP_IRB && o.irb.hwc)) ? 1 : 0;
    }
    auto hits = [](const std::vector<int> &a, const std::vector<int> &b) { for (int x : a) for (int y : b) if (x == y) return true; return false; };
    std::vector<std::vector<int>> deps(n), users(n);
    for (int i = (int)first; i < n; i++)
        for (int j = (int)first; j < i; j++)
            if (hits(wr[j], rd[i]) || hits(rd[j], wr[i]) || (hits(wr[j], wr[i]) && !(partial[i] && partial[j]))) { deps[i].push_back(j); users[j].push_back(i); }
    std::vector<int> prio(n, 1), heir(n, -1);                                   // longest remaining path in steps; the dependent that path runs through
    for (int i = n - 1; i >= (int)first; i--)
        for (int u : users[i]) if (prio[u] + 1 > prio[i]) { prio[i] = prio[u] + 1; heir[i] = u; }
    lanes = std::max(1, std::min(lanes, 8));
    while ((int)h->fork_streams.size() < lanes - 1) { hipStream_t q; SGX_CHECK_HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking)); h->fork_streams.push_back(q); }
    std::vector<hipStream_t> L(lanes); L[0] = st; for (int l = 1; l < lanes; l++) L[l] = h->fork_streams[l - 1];
    std::vector<int> tail(lanes, -1), lane_of(n, 0); std::vector<char> started(lanes, 0); started[0] = 1;
    std::vector<hipEvent_t> done(n, nullptr);
    // every event the capture can need exists before it begins (no resource creation while the thread captures)
    std::vector<hipEvent_t> pool((size_t)n + 2 * lanes); size_t next_ev = 0;
    for (hipEvent_t &e : pool) { SGX_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->fork_events.push_back(e); }
    SGX_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    // a failure inside the capture must not leave the caller's stream capturing
#define CAP_CHECK(x) do { if ((x) != hipSuccess) { hipGraph_t g2 = nullptr; (void)hipStreamEndCapture(st, &g2); if (g2) (void)hipGraphDestroy(g2); return SGX_ERR_DEVICE; } } while (0)
    for (int i = (int)first; i < n; i++) {
        int lane = -1;
        for (int j : deps[i]) if (tail[lane_of[j]] == j && heir[j] == i) { lane = lane_of[j]; break; }      // the critical successor keeps its producer's lane
        if (lane < 0 && !deps[i].empty()) {
            for (int l = 1; l < lanes && lane < 0; l++) {                                                       // a side lane whose tail has nothing critical left to wait for
                const int t = tail[l];
                if (t < 0 || heir[t] < 0 || heir[t] < i) lane = l;
            }
            if (lane < 0) for (int j : deps[i]) if (tail[lane_of[j]] == j) { lane = lane_of[j]; break; }      // no free lane: behind one of its producers
        }
        if (lane < 0) lane = 0;
        for (int j : deps[i]) if (lane_of[j] != lane) { CAP_CHECK(hipStreamWaitEvent(L[lane], done[j], 0)); started[lane] = 1; }
        if (!started[lane]) {                                                                                    // a side lane must enter the capture through an event of the origin stream
            hipEvent_t e = pool[next_ev++];
            CAP_CHECK(hipEventRecord(e, L[0])); CAP_CHECK(hipStreamWaitEvent(L[lane], e, 0)); started[lane] = 1;
        }
        run_op(h, h->ops[i], batch, L[lane]);
        lane_of[i] = lane; tail[lane] = i;
        if (!users[i].empty()) {                                                                                 // an event behind every step with dependents: the edge a later step on another lane waits on
            hipEvent_t e = pool[next_ev++]; CAP_CHECK(hipEventRecord(e, L[lane])); done[i] = e;
        }
    }
    for (int l = 1; l < lanes; l++) if (started[l] && tail[l] >= 0) {                                            // join: every side lane's tail before the capture ends on lane 0
        hipEvent_t e = pool[next_ev++];
        CAP_CHECK(hipEventRecord(e, L[l])); CAP_CHECK(hipStreamWaitEvent(L[0], e, 0));
    }
#undef CAP_CHECK
    SGX_CHECK_HIP(hipStreamEndCapture(st, graph));
    return SGX_OK;
}
#endif

// Batched forward from device-resident interleaved 3-channel u8 images (B x H x W x 3, row pitch in bytes).
// Leaves loc (num_priors*4) and softmax conf (num_priors*num_class) per image in device memory; returns their pointers.
extern "C" int sgx_det_forward_batch_dev(sgx_det *h, const uint8_t *d_img, int pitch, int batch, const float **d_loc, const float **d_conf, void *stream_)
{
    if (!h || !d_img || batch < 1 || batch > h->max_batch || pitch < ((3 * h->W + 3) & ~3) || (pitch & 3) || ((uintptr_t)d_img & 3)) return SGX_ERR_INVALID;   // rows are read as aligned dwords
    sgx_stream_t st = (sgx_stream_t)stream_;
    sgx_prof_begin(SGX_K_DET_FWD, st);
    run_first_step(h, d_img, pitch, batch, st);
    const size_t first = h->pre_fused ? 1 : 0;
#ifndef SGX_EMU
    // The plan after pre-processing only touches the handle's own blobs, so it is captured once per batch size into a hipGraph and replayed
    // (needs a non-default stream; SGX_DET_NO_GRAPH=1 or the legacy stream falls back to individual launches).
    static const bool no_graph = sgx_getenv("SGX_DET_NO_GRAPH") != nullptr;
    if (st != nullptr && !no_graph) {
        auto it = h->graphs.find(batch);
        if (it == h->graphs.end()) {
            hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
            static const int fork_lanes = sgx_getenv("SGX_DET_FORK") ? atoi(sgx_getenv("SGX_DET_FORK")) : SGX_DET_FORK_LANES;      // A/B tap: 1 = the one-stream chain
            if (fork_lanes > 1) { const int rc = capture_forked(h, first, batch, st, fork_lanes, &graph); if (rc != SGX_OK) return rc; }
            else {
                SGX_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (size_t i = first; i < h->ops.size(); i++) run_op(h, h->ops[i], batch, st);
                SGX_CHECK_HIP(hipStreamEndCapture(st, &graph));
            }
            // (tap) several executable instances used in turn: measured with the forked capture, whose launches block the host on this runtime — see SGX_DET_FORK_LANES above
            static const int nexec_env = sgx_getenv("SGX_DET_EXECS") ? atoi(sgx_getenv("SGX_DET_EXECS")) : SGX_DET_GRAPH_EXECS;
            std::vector<hipGraphExec_t> execs;
            for (int e = 0; e < std::max(1, std::min(nexec_env, 4)); e++) { SGX_CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)); execs.push_back(exec); }
            (void)hipGraphDestroy(graph);
            it = h->graphs.emplace(batch, execs).first;
        }
        { unsigned &turn = h->graph_turn[batch]; SGX_CHECK_HIP(hipGraphLaunch(it->second[turn % it->second.size()], st)); turn++; }
    } else
#endif
    for (size_t i = first; i < h->ops.size(); i++) run_op(h, h->ops[i], batch, st);
    sgx_prof_end(SGX_K_DET_FWD, st);
    SGX_CHECK_HIP(hipGetLastError());
    if (d_loc) *d_loc = h->blobs[h->loc_blob].d;
    if (d_conf) *d_conf = h->blobs[h->conf_blob].d;
    return SGX_OK;
}

// test / tuning taps: per-launch HIP-event time of every step of the plan (ms[0] = pre-processing, ms[1 + i] = op i), and a description of op i
SGX_TAP int sgx_det_debug_time_ops(sgx_det *h, const uint8_t *d_img, int pitch, int batch, int reps, float *ms, int cap, int *nops)
{
    if (!h || !d_img || !ms || !nops || batch < 1 || batch > h->max_batch || reps < 1) return SGX_ERR_INVALID;
    const int skip = h->pre_fused;                                  // step 0 is then the stem itself; step i >= 1 is ops[i]
    *nops = (int)h->ops.size() + 1 - skip;
    if (cap < *nops) return SGX_ERR_INVALID;
#ifndef SGX_EMU
    hipEvent_t a, b; SGX_CHECK_HIP(hipEventCreate(&a)); SGX_CHECK_HIP(hipEventCreate(&b));
    for (int i = 0; i < *nops; i++) {
        float tot = 0.f;
        for (int r = 0; r < reps + 1; r++) {
            SGX_CHECK_HIP(hipEventRecord(a, 0));
            if (i == 0) run_first_step(h, d_img, pitch, batch, 0); else run_op(h, h->ops[i - 1 + skip], batch, 0);
            SGX_CHECK_HIP(hipEventRecord(b, 0)); SGX_CHECK_HIP(hipEventSynchronize(b));
            float t = 0.f; SGX_CHECK_HIP(hipEventElapsedTime(&t, a, b)); if (r) tot += t;
        }
        ms[i] = tot / reps;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return SGX_OK;
#else
    return SGX_ERR_UNSUPPORTED;
#endif
}

// plan step i launched `reps` times on `stream` (asynchronous): a co-runner for interference experiments (round 6, tools/diag_lk_repeat.py)
SGX_TAP int sgx_det_debug_run_step(sgx_det *h, const uint8_t *d_img, int pitch, int batch, int i, int reps, void *stream)
{
    if (!h || !d_img || batch < 1 || batch > h->max_batch || reps < 1) return SGX_ERR_INVALID;
    const int skip = h->pre_fused, nops = (int)h->ops.size() + 1 - skip;
    if (i < 0 || i >= nops) return SGX_ERR_INVALID;
    for (int r = 0; r < reps; r++) { if (i == 0) run_first_step(h, d_img, pitch, batch, (sgx_stream_t)stream); else run_op(h, h->ops[i - 1 + skip], batch, (sgx_stream_t)stream); }
    return SGX_OK;
}

extern "C" int sgx_det_plan_step(const sgx_det *h, int i, char *buf, int cap)
{
    if (!h || !buf || cap < 16 || i < 0 || i > (int)h->ops.size() - h->pre_fused) return SGX_ERR_INVALID;
    if (i == 0 && !h->pre_fused) { snprintf(buf, cap, "preprocess %dx%d->%d", h->W, h->H, h->T); return SGX_OK; }
    if (i == 0) { const Op &s0 = h->ops[0]; snprintf(buf, cap, "kxk %s c%d->%d k%d s%d %dx%d->%dx%d epi%d pre %dx%d->%d", s0.name.c_str(), s0.inc, s0.outc, s0.k, s0.stride, s0.H, s0.W, s0.Ho, s0.Wo,
                                                     (int)s0.epi.size() + (s0.act ? 1 : 0), h->W, h->H, h->T); return SGX_OK; }
    const Op &o = h->ops[i - 1 + h->pre_fused];
    static const char *kn[] = { "pw", "kxk", "binary", "unary", "permute_into", "copy_into", "softmax", "block", "irb", "se_gate" };
    if (o.kind == OP_SE_GATE) { snprintf(buf, cap, "se_gate %s c%d->%d->%d %dx%d%s", o.name.c_str(), o.sg.Cout, o.sg.Cq, o.sg.Cout, o.H, o.W, o.sg_res_blob >= 0 ? " +res" : ""); return SGX_OK; }
    if (o.kind == OP_PW || o.kind == OP_KXK)
        snprintf(buf, cap, "%s %s c%d->%d k%d s%d %s%dx%d->%dx%d epi%d%s%s", kn[o.kind], o.name.c_str(), o.inc, o.outc, o.k, o.stride, o.depthwise ? "dw " : "", o.H, o.W,
                 o.kind == OP_PW ? o.H : o.Ho, o.kind == OP_PW ? o.W : o.Wo, (int)o.epi.size() + (o.act ? 1 : 0), o.hwc ? " hwc" : "", (o.kind == OP_PW && h->gemm == 1 && o.wS && o.inc >= (sgx_getenv("SGX_PW3_MINK") ? atoi(sgx_getenv("SGX_PW3_MINK")) : 64)) ? " bf16x3" : "");
    else if (o.kind == OP_FUSED_BLOCK)
        snprintf(buf, cap, "block %s c%d->%d->%d k%d s%d %dx%d->%dx%d tile %dx%d%s%s%s", o.name.c_str(), o.fb.Cin, o.fb.Cmid, o.fb.Cout, o.fb.K, o.fb.stride, o.H, o.W, o.Ho, o.Wo, o.fb.TOH, o.fb.TOW,
                 o.fb.Cq ? " se" : "", o.fb_res_blob >= 0 ? " +res" : "", o.fb.hrb ? (o.fb.hrb_occ == 4 ? " hrb4 bf16x3" : o.fb.hrb_occ == 3 ? " hrb3 bf16x3" : " hrb2 bf16x3") : "");
    else if (o.kind == OP_IRB)
        snprintf(buf, cap, "irb %s c%d->%d->%d q%d k%d s%d %dx%d->%dx%d G%d bands%d buf%d%s%s%s%s", o.name.c_str(), o.irb.Cin, o.irb.Cexp, o.irb.Cout, o.irb.Cq, o.irb.K, o.irb.S, o.H, o.W, o.Ho, o.Wo,
                 o.irb.G, o.irb.nbands, o.irb.nbuf, o.irb.has_expand ? "" : " noexp", o.irb_res_blob >= 0 ? " +res" : "", o.hwc ? (o.irb.Cout2 ? " hwc dual" : " hwc") : "", o.irb.gemm == 1 ? " bf16x3" : "");
    else snprintf(buf, cap, "%s %s n=%zu", kn[o.kind], o.name.c_str(), h->blobs[o.in0].n);
    return SGX_OK;
}

extern "C" int sgx_det_info(const sgx_det *h, int32_t *num_priors, int32_t *num_class, int32_t *num_kernels, double *gmac)
{
    if (!h) return SGX_ERR_INVALID;
    if (num_priors) *num_priors = h->num_priors; if (num_class) *num_class = h->num_class; if (num_kernels) *num_kernels = (int)h->ops.size() + 1 - h->pre_fused; if (gmac) *gmac = h->gmac;
    return SGX_OK;
}

// test tap: copy one image's blob (by ncnn blob name) to the host; returns element count in *n
SGX_TAP int sgx_det_debug_read_blob(sgx_det *h, const char *name, int image, float *dst, int cap, int *n)
{
    if (!h || !name || !n) return SGX_ERR_INVALID;
    auto it = h->blob_id.find(name); if (it == h->blob_id.end()) return SGX_ERR_INVALID;
    int id = it->second; while (h->blobs[id].alias >= 0) id = h->blobs[id].alias;
    const Blob &b = h->blobs[id];
    if (!b.d || image < 0 || image >= h->max_batch) return SGX_ERR_INVALID;
    *n = (int)b.n;
    if (dst && cap >= (int)b.n) SGX_CHECK_HIP(hipMemcpy(dst, b.d + (size_t)image * b.n, b.n * 4, hipMemcpyDeviceToHost));
    return SGX_OK;
}

static int run_detection_output(sgx_det *h, int batch, sgx_det_result *d_results, float *d_boxes, int32_t *d_nboxes, int max_boxes, int32_t *d_have_dynamic, sgx_stream_t st);

// Detector2D::detect, device-resident: forward, then ncnn DetectionOutput (decode + per-class NMS + keep_top_k) and the detect() filtering as two kernels
// (k_det_class_nms, k_det_merge).  d_results[batch] receives the same struct the host entry returns; d_boxes / d_nboxes / d_have_dynamic (optional) are
// the person rectangles, their count and the have-dynamic flag in the layout sgx_dynamic_mask_batch_dev and sgx_frame_compact_keys_batch_dev take.
extern "C" int sgx_det_detect_batch_dev(sgx_det *h, const uint8_t *d_img, int pitch, int batch, sgx_det_result *d_results,
                                        float *d_boxes, int32_t *d_nboxes, int max_boxes, int32_t *d_have_dynamic, void *stream_)
{
    if (!h || !d_results || ((d_boxes != nullptr) != (d_nboxes != nullptr)) || (d_boxes && max_boxes < 1)) return SGX_ERR_INVALID;
    const float *dl = nullptr, *dc = nullptr;
    int rc = sgx_det_forward_batch_dev(h, d_img, pitch, batch, &dl, &dc, stream_);
    if (rc != SGX_OK) return rc;
    return run_detection_output(h, batch, d_results, d_boxes, d_nboxes, max_boxes, d_have_dynamic, (sgx_stream_t)stream_);
}

static int run_detection_output(sgx_det *h, int batch, sgx_det_result *d_results, float *d_boxes, int32_t *d_nboxes, int max_boxes, int32_t *d_have_dynamic, sgx_stream_t st)
{
    SgxDetOut P; P.n = h->num_priors; P.nc = h->num_class; P.nms_top_k = h->nms_top_k; P.keep_top_k = h->keep_top_k; P.nms_th = h->nms_th; P.conf_th = h->conf_th;
    P.var0 = h->dvar[0]; P.var1 = h->dvar[1]; P.var2 = h->dvar[2]; P.var3 = h->dvar[3];
    { static const bool skip = sgx_getenv("SGX_DET_SKIP_OUTPUT") != nullptr; if (skip) return SGX_OK; }      // timing tap (round 6): what the per-class NMS chains cost the detector stream's critical path
    sgx_prof_begin(SGX_K_DET_OUT, st);
    SGX_LAUNCH_DYN(k_det_class_nms, dim3(h->num_class - 1, batch), dim3(256), (size_t)std::max(h->num_priors, 7 * SGX_DO_TOPK) * 4 + 16, st, P, h->blobs[h->loc_blob].d, h->blobs[h->conf_blob].d, h->d_priors, h->d_cls_rows, h->d_cls_count);
    SGX_LAUNCH(k_det_merge, dim3(batch), dim3(256), st, P, h->d_cls_rows, h->d_cls_count, h->det_th, h->dyn_th, h->W, h->H, h->T, d_results, d_boxes, d_nboxes, max_boxes, d_have_dynamic);
    sgx_prof_end(SGX_K_DET_OUT, st);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

// test tap: DetectionOutput + detect() filtering alone on caller-supplied head outputs (loc: batch x num_priors x 4, conf: batch x num_priors x num_class)
SGX_TAP int sgx_det_debug_detection_output(sgx_det *h, const float *loc, const float *conf, int batch, sgx_det_result *results)
{
    if (!h || !loc || !conf || !results || batch < 1 || batch > h->max_batch) return SGX_ERR_INVALID;
    SGX_CHECK_HIP(hipMemcpy(h->blobs[h->loc_blob].d, loc, sizeof(float) * 4 * (size_t)h->num_priors * batch, hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(h->blobs[h->conf_blob].d, conf, sizeof(float) * (size_t)h->num_priors * h->num_class * batch, hipMemcpyHostToDevice));
    int rc = run_detection_output(h, batch, h->d_results, nullptr, nullptr, 0, nullptr, (sgx_stream_t)0);
    if (rc != SGX_OK) return rc;
    SGX_CHECK_HIP(hipMemcpy(results, h->d_results, sizeof(sgx_det_result) * batch, hipMemcpyDeviceToHost));
    return SGX_OK;
}

// Detector2D::detect for `batch` host images (interleaved 3-channel u8, as cv::Mat bgr.data).  Per image: raw detection_out
// rows (<= keep_top_k), the filtered objects and the "person" rectangles for mapping / for the dynamic-feature mask.
extern "C" int sgx_det_detect(sgx_det *h, const uint8_t *images, int pitch, int batch, sgx_det_result *results)
{
    if (!h || !images || !results || batch < 1 || batch > h->max_batch || pitch < 3 * h->W) return SGX_ERR_INVALID;
    const int ipitch = (3 * h->W + 3) & ~3;                      // device copy: rows padded to whole dwords
    for (int b = 0; b < batch; b++)
        for (int y = 0; y < h->H; y++)
            SGX_CHECK_HIP(hipMemcpyAsync(h->d_img + ((size_t)b * h->H + y) * ipitch, images + ((size_t)b * h->H + y) * pitch, (size_t)3 * h->W, hipMemcpyHostToDevice, 0));
    int rc = sgx_det_detect_batch_dev(h, h->d_img, ipitch, batch, h->d_results, nullptr, nullptr, 0, nullptr, nullptr);
    if (rc != SGX_OK) return rc;
    SGX_CHECK_HIP(hipMemcpy(results, h->d_results, sizeof(sgx_det_result) * batch, hipMemcpyDeviceToHost));
    return SGX_OK;
}

extern "C" int sgx_dynamic_mask_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const int32_t *d_n, const float *d_prev_xy, const double *d_F,
                                          const float *d_boxes, const int32_t *d_nboxes, int max_boxes, uint8_t *d_keep, void *stream)
{
    if (batch < 1 || cap < 1 || !d_keys || !d_n || !d_prev_xy || !d_F || !d_boxes || !d_nboxes || !d_keep || max_boxes < 0) return SGX_ERR_INVALID;
    sgx_prof_begin(SGX_K_MASK, (sgx_stream_t)stream);
    SGX_LAUNCH(k_dynamic_mask, dim3((cap + 255) / 256, batch), dim3(256), (sgx_stream_t)stream, cap, (const uint8_t *)d_keys, d_n, d_prev_xy, d_F, d_boxes, d_nboxes, max_boxes, d_keep);
    sgx_prof_end(SGX_K_MASK, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_frame_compact_keys_batch_dev(int batch, int cap, const sgx_keypoint *d_keys, const uint8_t *d_desc, const int32_t *d_n, const uint8_t *d_keep,
                                                const int32_t *d_have_dynamic, int nfeatures, sgx_keypoint *d_keys_out, uint8_t *d_desc_out, int32_t *d_n_out, void *stream)
{
    if (batch < 1 || cap < 1 || !d_keys || !d_desc || !d_n || !d_keep || !d_keys_out || !d_desc_out || !d_n_out || (const void *)d_keys == (const void *)d_keys_out) return SGX_ERR_INVALID;
    sgx_prof_begin(SGX_K_MASK, (sgx_stream_t)stream);
    SGX_LAUNCH(k_compact_keys, dim3(batch), dim3(256), (sgx_stream_t)stream, cap, (const uint8_t *)d_keys, d_desc, d_n, d_keep, d_have_dynamic, (float)nfeatures * 0.1f,
               (uint8_t *)d_keys_out, d_desc_out, d_n_out);
    sgx_prof_end(SGX_K_MASK, (sgx_stream_t)stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}
