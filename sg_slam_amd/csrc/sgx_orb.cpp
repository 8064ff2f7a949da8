// sgx_orb.cpp — host side of the ORB extractor C-ABI (include/sgx.h): handle, tables, launches.
// Compiled as HIP for gfx950 (product) or as plain C++ with -DSGX_EMU (kernel-logic emulator,
// tests only).  Reference behaviour: src/sg-slam/src/ORBextractor.cc (cited inline).
#include "sgx_orb_kernels.h"
#include "sgx_prof.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include "../../include/sgx_orb_pattern.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#ifdef SGX_EMU
thread_local sgx_dim3 blockIdx, blockDim, gridDim;
#endif

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

struct sgx_orb {
    sgx_orb_config cfg;
    SgxOrbGeom g;
    float scale[SGX_MAX_LEVELS], inv_scale[SGX_MAX_LEVELS], sigma2[SGX_MAX_LEVELS], inv_sigma2[SGX_MAX_LEVELS];
    int umax_h[16];
    // device tables
    SgxCell *d_cells = nullptr;
    SgxXTab *d_xt[SGX_MAX_LEVELS] = {};
    SgxYTab *d_yt[SGX_MAX_LEVELS] = {};
    // fused pyramid (k_pyramid): concatenated tables, per (tile, level) rects
    SgxXTab *d_xt_all = nullptr; SgxYTab *d_yt_all = nullptr; SgxPyrRect *d_pyr_rects = nullptr; SgxPyrTabs pyr_tabs; int pyr_tiles = 0, pyr_lds = 0;
    int *d_umax = nullptr;
    signed char *d_pattern = nullptr;
    // device workspace (sized for cfg.max_batch)
    uint8_t *d_pyr = nullptr;
    uint8_t *d_blur = nullptr; SgxBlurTile *d_blur_tiles = nullptr;     // blurred copy of every level (k_blur_levels) and its tile table
    uint32_t *d_cand = nullptr;
    int *d_cand_count = nullptr;
    uint16_t *d_node_scratch = nullptr;
    uint32_t *d_sel = nullptr;
    int *d_sel_count = nullptr;
    uint32_t *d_status = nullptr;
    // single-frame staging for sgx_orb_extract
    uint8_t *d_gray1 = nullptr; uint8_t *d_kps1 = nullptr; uint8_t *d_desc1 = nullptr; int *d_count1 = nullptr;
    int last_batch = 0;
    int oct_maxlim = 0;            // largest octree list capacity over the levels (quota + 3 or 4*nIni)
};

static thread_local int g_orb_unfused_pyramid = 0;      // test tap: 1 = one k_resize launch per level instead of the fused k_pyramid
SGX_TAP int sgx_orb_debug_set_unfused_pyramid(int on) { g_orb_unfused_pyramid = on ? 1 : 0; return SGX_OK; }

static inline int cvround_f(float v) { return (int)lrintf(v); }
static inline int cvround_d(double v) { return (int)lrint(v); }

// ORBextractor::ORBextractor, ORBextractor.cc:411-471 (scale tables, per-level quotas, umax)
static void build_tables(sgx_orb *h)
{
    const sgx_orb_config &c = h->cfg;
    const double sf = (double)c.scale_factor;           // member `double scaleFactor` (ORBextractor.h:99) set from a float
    const int nl = c.nlevels;
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) { h->scale[i] = (float)((double)h->scale[i - 1] * sf); h->sigma2[i] = h->scale[i] * h->scale[i]; }
    for (int i = 0; i < nl; i++) { h->inv_scale[i] = 1.0f / h->scale[i]; h->inv_sigma2[i] = 1.0f / h->sigma2[i]; }
    const float factor = (float)(1.0f / sf);
    float want = c.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) { h->g.lv[l].quota = cvround_f(want); sum += h->g.lv[l].quota; want *= factor; }
    h->g.lv[nl - 1].quota = c.nfeatures - sum > 0 ? c.nfeatures - sum : 0;
    // umax (:455-470)
    int umax[17] = {0};
    const int HP = 15;
    const int vmax = (int)floor(HP * sqrtf(2.f) / 2 + 1), vmin = (int)ceil(HP * sqrtf(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = cvround_d(sqrt((double)HP * HP - v * v));
    for (int v = HP, v0 = 0; v >= vmin; --v) { while (umax[v0] == umax[v0 + 1]) ++v0; umax[v] = v0; ++v0; }
    for (int i = 0; i < 16; i++) h->umax_h[i] = umax[i];
}

// cv::resize coefficient tables (OpenCV 3.4 imgproc/resize.cpp, INTER_LINEAR, 8U fixed point)
static void build_resize_tables(int sw, int sh, int dw, int dh, std::vector<SgxXTab> &xt, std::vector<SgxYTab> &yt)
{
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    xt.resize(dw); yt.resize(dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xt[dx].sx = (short)sx; xt[dx].sx1 = (short)(sx + 1 < sw ? sx + 1 : sx);
        xt[dx].a0 = (short)cvround_f((1.f - fx) * 2048); xt[dx].a1 = (short)cvround_f(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        int r0 = sy, r1 = sy + 1;
        if (r0 < 0) r0 = 0; if (r0 > sh - 1) r0 = sh - 1;
        if (r1 < 0) r1 = 0; if (r1 > sh - 1) r1 = sh - 1;
        yt[dy].sy0 = (short)r0; yt[dy].sy1 = (short)r1;
        yt[dy].b0 = (short)cvround_f((1.f - fy) * 2048); yt[dy].b1 = (short)cvround_f(fy * 2048);
    }
}

extern "C" const char *sgx_version(void) {
#ifdef SGX_EMU
    return "sgx 0.1 (EMULATOR - tests only)";
#else
    return "sgx 0.1 (gfx950)";
#endif
}

extern "C" const char *sgx_status_string(int s)
{
    switch (s) {
    case SGX_OK: return "ok";
    case SGX_ERR_INVALID: return "invalid argument";
    case SGX_ERR_UNSUPPORTED: return "unsupported geometry";
    case SGX_ERR_NOMEM: return "out of memory";
    case SGX_ERR_DEVICE: return "HIP runtime error";
    case SGX_ERR_OVERFLOW: return "device buffer overflow";
    default: return "unknown";
    }
}

extern "C" void sgx_orb_destroy(sgx_orb *h)
{
    if (!h) return;
    (void)hipFree(h->d_cells); (void)hipFree(h->d_umax); (void)hipFree(h->d_pattern); (void)hipFree(h->d_pyr); (void)hipFree(h->d_cand);
    (void)hipFree(h->d_cand_count); (void)hipFree(h->d_node_scratch); (void)hipFree(h->d_sel); (void)hipFree(h->d_sel_count); (void)hipFree(h->d_status);
    (void)hipFree(h->d_gray1); (void)hipFree(h->d_kps1); (void)hipFree(h->d_desc1); (void)hipFree(h->d_count1);
    for (int l = 0; l < SGX_MAX_LEVELS; l++) { (void)hipFree(h->d_xt[l]); (void)hipFree(h->d_yt[l]); }
    (void)hipFree(h->d_xt_all); (void)hipFree(h->d_yt_all); (void)hipFree(h->d_pyr_rects); (void)hipFree(h->d_blur); (void)hipFree(h->d_blur_tiles);
    delete h;
}

extern "C" int sgx_orb_create(const sgx_orb_config *cfg, sgx_orb **out)
{
    if (!cfg || !out) return SGX_ERR_INVALID;
    if (cfg->nlevels < 1 || cfg->nlevels > SGX_MAX_LEVELS || cfg->nfeatures < 1 || cfg->width < 64 || cfg->height < 64 ||
        cfg->width > 4000 || cfg->height > 4000 || cfg->max_batch < 1 || !(cfg->scale_factor > 1.0f)) return SGX_ERR_INVALID;
    sgx_orb *h = new sgx_orb();
    h->cfg = *cfg;
    SgxOrbGeom &g = h->g;
    memset(&g, 0, sizeof g);
    g.nlevels = cfg->nlevels; g.W = cfg->width; g.H = cfg->height; g.ini_th = cfg->ini_th_fast; g.min_th = cfg->min_th_fast;
    build_tables(h);
    std::vector<SgxCell> cells;
    int off = 0, kp_cap = 0, cand_off = 0;
    for (int l = 0; l < g.nlevels; l++) {
        SgxLevel &L = g.lv[l];
        L.w = cvround_f((float)cfg->width * h->inv_scale[l]);      // ORBextractor.cc:1112-1113
        L.h = cvround_f((float)cfg->height * h->inv_scale[l]);
        L.stride = (L.w + 63) & ~63;
        L.scale = h->scale[l];
        L.patch_size = (int)(31 * h->scale[l]);                     // :838
        if (l > 0) { L.off = off; off += L.stride * L.h; }
        if (L.w < 2 * SGX_EDGE + 8 || L.h < 2 * SGX_EDGE + 8) { sgx_orb_destroy(h); return SGX_ERR_UNSUPPORTED; }
        // FAST cell grid, ORBextractor.cc:774-808
        const float W = 30;
        const int minBX = SGX_BORDER, minBY = SGX_BORDER, maxBX = L.w - SGX_EDGE + 3, maxBY = L.h - SGX_EDGE + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        L.ncols = (int)(width / W); L.nrows = (int)(height / W);
        if (L.ncols < 1 || L.nrows < 1) { sgx_orb_destroy(h); return SGX_ERR_UNSUPPORTED; }
        L.wcell = (int)ceilf(width / L.ncols); L.hcell = (int)ceilf(height / L.nrows);
        L.cell0 = (int)cells.size();
        L.cand_off = cand_off; L.cand_cap = 0;
        if (L.wcell + 6 > SGX_TILE_MAX || L.hcell + 6 > SGX_TILE_MAX || L.ncols > 1023 || L.nrows > 1023 || L.wcell > 1023 || L.hcell > 1023) {
            sgx_orb_destroy(h); return SGX_ERR_UNSUPPORTED; }
        for (int i = 0; i < L.nrows; i++) {
            const float iniY = (float)(minBY + i * L.hcell);
            float maxY = iniY + L.hcell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < L.ncols; j++) {
                const float iniX = (float)(minBX + j * L.wcell);
                float maxX = iniX + L.wcell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                SgxCell c; c.level = (short)l; c.x0 = (short)iniX; c.y0 = (short)iniY;
                c.cw = (short)((int)maxX - (int)iniX); c.ch = (short)((int)maxY - (int)iniY);
                c.ox = (short)(j * L.wcell); c.oy = (short)(i * L.hcell);
                { const int ng = ((c.x0 & 3) + c.cw + 3) >> 2; c.pad = (short)(unsigned short)((65536 + ng - 1) / ng); }      // ceil(2^16 / ng): k_fast_cells divides its task index by ng (ng >= 2, tasks < 4096: exact)
                if (c.cw >= 7 && c.ch >= 7) {                        // cv::FAST yields nothing on tiles < 7 px
                    cells.push_back(c);
                    L.cand_cap += ((c.cw - 6 + 1) / 2) * ((c.ch - 6 + 1) / 2);   // strict-'>' NMS: no two survivors are 8-adjacent
                }
            }
        }
        const int nIni = (int)roundf((float)(maxBX - minBX) / (float)(maxBY - minBY));
        // a level whose bordered area is more than twice as tall as wide has nIni = 0 root nodes: the reference divides by it and indexes an empty vector
        // (ORBextractor.cc:544-566, undefined behaviour) — refuse such geometries instead of returning a level without keypoints
        if (nIni < 1 || nIni > 64) { sgx_orb_destroy(h); return SGX_ERR_UNSUPPORTED; }
        int lim = L.quota + 3; if (lim < 4 * nIni) lim = 4 * nIni;
        if (lim > SGX_OCT_MAXN) { sgx_orb_destroy(h); return SGX_ERR_UNSUPPORTED; }
        kp_cap += lim;
        if (lim > h->oct_maxlim) h->oct_maxlim = lim;
        L.cand_cap = (L.cand_cap + 63) & ~63;
        cand_off += L.cand_cap;
    }
    g.cand_pitch = cand_off;
    {   // k_fast_cells LDS carve for the largest tile of this geometry
        int mw = 7, mh = 7;
        for (const SgxCell &cc : cells) { if (cc.cw > mw) mw = cc.cw; if (cc.ch > mh) mh = cc.ch; }
        const int tile_b = mh * SGX_TILE_STRIDE, q_b = ((mw - 6) * (mh - 6) * 2 + 15) & ~15, o_b = (((mw - 5) / 2) * ((mh - 5) / 2) * 4 + 15) & ~15;
        g.fast_off_score = tile_b; g.fast_off_qlist = 2 * tile_b; g.fast_off_out = 2 * tile_b + q_b; g.fast_lds_bytes = 2 * tile_b + q_b + o_b;
    }
    g.pyr_pitch = (off + 255) & ~255;
    std::vector<SgxBlurTile> btiles;
    {
        int boff = 0;
        for (int l = 0; l < g.nlevels; l++) {
            SgxLevel &L = g.lv[l];
            L.bstride = (L.w + 63) & ~63; L.boff = boff; boff += L.bstride * L.h;
            for (int y0 = 0; y0 < L.h; y0 += SGX_BT_H) for (int x0 = 0; x0 < L.w; x0 += SGX_BT_W) {
                SgxBlurTile t; memset(&t, 0, sizeof t);
                t.level = (short)l; t.x0 = (short)x0; t.y0 = (short)y0; t.w = (short)std::min(SGX_BT_W, L.w - x0); t.h = (short)std::min(SGX_BT_H, L.h - y0);
                btiles.push_back(t);
            }
        }
        g.blur_pitch = (boff + 255) & ~255; g.nblur_tiles = (int)btiles.size();
    }
    g.ncells = (int)cells.size();
    g.kp_cap = kp_cap;

    const int B = cfg->max_batch, nl = g.nlevels;
#define SGX_ALLOC(p, bytes) do { if (hipMalloc((void **)&(p), (bytes)) != hipSuccess) { sgx_orb_destroy(h); return SGX_ERR_NOMEM; } } while (0)
    SGX_ALLOC(h->d_cells, cells.size() * sizeof(SgxCell));
    SGX_ALLOC(h->d_umax, 16 * sizeof(int));
    SGX_ALLOC(h->d_pattern, 1024);
    SGX_ALLOC(h->d_pyr, (size_t)B * g.pyr_pitch + 256);
    SGX_ALLOC(h->d_blur, (size_t)B * g.blur_pitch + 256);
    SGX_ALLOC(h->d_blur_tiles, btiles.size() * sizeof(SgxBlurTile));
    SGX_CHECK_HIP(hipMemcpy(h->d_blur_tiles, btiles.data(), btiles.size() * sizeof(SgxBlurTile), hipMemcpyHostToDevice));
    SGX_ALLOC(h->d_cand, (size_t)B * g.cand_pitch * 4);
    SGX_ALLOC(h->d_node_scratch, (size_t)B * g.cand_pitch * 2);
    SGX_ALLOC(h->d_cand_count, (size_t)B * nl * 4);
    SGX_ALLOC(h->d_sel, (size_t)B * nl * SGX_OCT_MAXN * 4);
    SGX_ALLOC(h->d_sel_count, (size_t)B * nl * 4);
    SGX_ALLOC(h->d_status, 4);
    SGX_ALLOC(h->d_gray1, (size_t)((g.W + 3) & ~3) * g.H);             // staging copy of a host frame, rows padded to dwords (the kernels stage aligned dwords)
    SGX_ALLOC(h->d_kps1, (size_t)kp_cap * 28);
    SGX_ALLOC(h->d_desc1, (size_t)kp_cap * 32);
    SGX_ALLOC(h->d_count1, 4);
    SGX_CHECK_HIP(hipMemcpyAsync(h->d_cells, cells.data(), cells.size() * sizeof(SgxCell), hipMemcpyHostToDevice, 0));
    SGX_CHECK_HIP(hipMemcpyAsync(h->d_umax, h->umax_h, 16 * sizeof(int), hipMemcpyHostToDevice, 0));
    SGX_CHECK_HIP(hipMemcpyAsync(h->d_pattern, sgx_orb_pattern_xy, 1024, hipMemcpyHostToDevice, 0));
    SGX_CHECK_HIP(hipMemsetAsync(h->d_status, 0, 4, 0));
    for (int l = 1; l < nl; l++) {
        std::vector<SgxXTab> xt; std::vector<SgxYTab> yt;
        build_resize_tables(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h, xt, yt);
        SGX_ALLOC(h->d_xt[l], xt.size() * sizeof(SgxXTab));
        SGX_ALLOC(h->d_yt[l], yt.size() * sizeof(SgxYTab));
        SGX_CHECK_HIP(hipMemcpyAsync(h->d_xt[l], xt.data(), xt.size() * sizeof(SgxXTab), hipMemcpyHostToDevice, 0));
        SGX_CHECK_HIP(hipMemcpyAsync(h->d_yt[l], yt.data(), yt.size() * sizeof(SgxYTab), hipMemcpyHostToDevice, 0));
        SGX_CHECK_HIP(hipStreamSynchronize(0));   // xt/yt are stack vectors
    }
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    if (nl > 1) {   // ---- k_pyramid plan: tiles are a fixed partition of every level; needed regions are propagated from the coarsest level up
        std::vector<std::vector<SgxXTab>> xts(nl); std::vector<std::vector<SgxYTab>> yts(nl);
        std::vector<SgxXTab> xall; std::vector<SgxYTab> yall;
        memset(&h->pyr_tabs, 0, sizeof h->pyr_tabs);
        for (int l = 1; l < nl; l++) {
            build_resize_tables(g.lv[l - 1].w, g.lv[l - 1].h, g.lv[l].w, g.lv[l].h, xts[l], yts[l]);
            h->pyr_tabs.xoff[l] = (int)xall.size(); h->pyr_tabs.yoff[l] = (int)yall.size();
            xall.insert(xall.end(), xts[l].begin(), xts[l].end()); yall.insert(yall.end(), yts[l].begin(), yts[l].end());
        }
        const int T = 32, ntx = (g.lv[nl - 1].w + T - 1) / T, nty = (g.lv[nl - 1].h + T - 1) / T;
        std::vector<SgxPyrRect> rects((size_t)ntx * nty * nl);
        int max_a = 0, max_b = 0, max_x = 0, max_y = 0;
        for (int j = 0; j < nty; j++) for (int i = 0; i < ntx; i++) {
            SgxPyrRect *R = &rects[((size_t)j * ntx + i) * nl];
            int nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;                 // needed region of level l+1 (exclusive ends)
            for (int l = nl - 1; l >= 0; l--) {
                const int Wl = g.lv[l].w, Hl = g.lv[l].h;
                int ax0 = Wl, ax1 = 0, ay0 = Hl, ay1 = 0;
                if (l >= 1) {                                        // owned part of this level
                    ax0 = i == 0 ? 0 : (int)(((long)i * Wl / ntx) & ~3L); ax1 = i == ntx - 1 ? Wl : (int)(((long)(i + 1) * Wl / ntx) & ~3L);
                    ay0 = (int)((long)j * Hl / nty); ay1 = j == nty - 1 ? Hl : (int)((long)(j + 1) * Hl / nty);
                }
                SgxPyrRect &r = R[l];
                r.ox0 = (short)ax0; r.ox1 = (short)ax1; r.oy0 = (short)ay0; r.oy1 = (short)ay1;
                if (l < nl - 1) {                                    // sources of the next level's needed region
                    for (int x = nx0; x < std::min(nx1, g.lv[l + 1].w); x++) { ax0 = std::min(ax0, (int)xts[l + 1][x].sx); ax1 = std::max(ax1, (int)xts[l + 1][x].sx1 + 1); }
                    for (int y = ny0; y < ny1; y++) { ay0 = std::min(ay0, (int)yts[l + 1][y].sy0); ay1 = std::max(ay1, (int)yts[l + 1][y].sy1 + 1); }
                }
                ax0 &= ~3; ax1 = (ax1 + 3) & ~3;                     // dword groups; groups past the image width are computed as zeros / read inside the row pitch
                r.nx0 = (short)ax0; r.ny0 = (short)ay0; r.nw = (short)(ax1 - ax0); r.nh = (short)(ay1 - ay0);
                const int q = r.nw / 4; r.qmagic = q <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)q - 1) / (unsigned)q); r.pad = 0;
                const int bytes = (int)r.nw * r.nh;
                if (l & 1) max_b = std::max(max_b, bytes); else max_a = std::max(max_a, bytes);
                nx0 = ax0; nx1 = ax1; ny0 = ay0; ny1 = ay1;
            }
            int xo = 0, yo = 0;
            for (int l = 1; l < nl; l++) { R[l].xo = xo; R[l].yo = yo; xo += R[l].nw; yo += R[l].nh; }
            R[0].xo = R[0].yo = 0;
            max_x = std::max(max_x, xo); max_y = std::max(max_y, yo);
        }
        h->pyr_tabs.lds_a = (max_a + 15) & ~15; h->pyr_tabs.lds_b = (max_b + 15) & ~15; h->pyr_tabs.lds_x = max_x * (int)sizeof(SgxXTab);
        h->pyr_lds = h->pyr_tabs.lds_a + h->pyr_tabs.lds_b + h->pyr_tabs.lds_x + max_y * (int)sizeof(SgxYTab);
        h->pyr_tiles = ntx * nty;
        if (h->pyr_lds > 64 * 1024) h->pyr_tiles = 0;               // geometry too large for the fused plan: per-level k_resize launches
        SGX_ALLOC(h->d_xt_all, xall.size() * sizeof(SgxXTab)); SGX_ALLOC(h->d_yt_all, yall.size() * sizeof(SgxYTab)); SGX_ALLOC(h->d_pyr_rects, rects.size() * sizeof(SgxPyrRect));
        SGX_CHECK_HIP(hipMemcpy(h->d_xt_all, xall.data(), xall.size() * sizeof(SgxXTab), hipMemcpyHostToDevice));
        SGX_CHECK_HIP(hipMemcpy(h->d_yt_all, yall.data(), yall.size() * sizeof(SgxYTab), hipMemcpyHostToDevice));
        SGX_CHECK_HIP(hipMemcpy(h->d_pyr_rects, rects.data(), rects.size() * sizeof(SgxPyrRect), hipMemcpyHostToDevice));
    }
#undef SGX_ALLOC
    *out = h;
    return SGX_OK;
}

extern "C" int sgx_orb_keypoint_capacity(const sgx_orb *h) { return h ? h->g.kp_cap : SGX_ERR_INVALID; }

extern "C" int sgx_orb_get_tables(const sgx_orb *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2, int32_t *per_level)
{
    if (!h) return SGX_ERR_INVALID;
    for (int l = 0; l < h->g.nlevels; l++) {
        if (scale) scale[l] = h->scale[l];
        if (inv_scale) inv_scale[l] = h->inv_scale[l];
        if (sigma2) sigma2[l] = h->sigma2[l];
        if (inv_sigma2) inv_sigma2[l] = h->inv_sigma2[l];
        if (per_level) per_level[l] = h->g.lv[l].quota;
    }
    return SGX_OK;
}

// k_octree launches.  Candidate-count classes: small LDS footprints let several (frame, level) workgroups share a CU; each block runs in exactly one launch.
static void launch_octree(sgx_orb *h, int batch, sgx_stream_t stream)
{
    const SgxOrbGeom &g = h->g; const int nl = g.nlevels;
    static const int oct_threads = sgx_getenv("SGX_TUNE_OCT_THREADS") ? atoi(sgx_getenv("SGX_TUNE_OCT_THREADS")) : SGX_OCT_THREADS;   // env = tuning tap (64..SGX_OCT_THREADS)
    // (levels, frames) dispatch order.  The (frames, levels) order — all level-0 workgroups first, small levels in the tail; tap below — runs the kernel itself
    // 30 % faster at 256 frames (0.33 -> 0.22 ms) but the two-stream pipeline 1.5 % slower (A/B on one box: 112.5 k vs 114.1 k frames/s), so it is not the default
    static const bool frame_fast = sgx_getenv("SGX_TUNE_OCT_FRAME_FAST") != nullptr;
    const dim3 ogrid = frame_fast ? dim3(batch, nl) : dim3(nl, batch);
    if (h->oct_maxlim <= 256) {
        auto ka = k_octree<true, 256, 2048>; auto kb = k_octree<true, 256, SGX_CAND_LDS>; auto kc = k_octree<false, 256, 1>;
        SGX_LAUNCH(ka, ogrid, dim3(oct_threads), stream, g, h->d_cand, h->d_cand_count, h->d_node_scratch, h->d_sel, h->d_sel_count, h->d_status, -1);
        SGX_LAUNCH(kb, ogrid, dim3(oct_threads), stream, g, h->d_cand, h->d_cand_count, h->d_node_scratch, h->d_sel, h->d_sel_count, h->d_status, 2048);
        SGX_LAUNCH(kc, ogrid, dim3(oct_threads), stream, g, h->d_cand, h->d_cand_count, h->d_node_scratch, h->d_sel, h->d_sel_count, h->d_status, SGX_CAND_LDS);
    } else {
        auto kb = k_octree<true, SGX_OCT_MAXN, SGX_CAND_LDS>; auto kc = k_octree<false, SGX_OCT_MAXN, 1>;
        SGX_LAUNCH(kb, ogrid, dim3(oct_threads), stream, g, h->d_cand, h->d_cand_count, h->d_node_scratch, h->d_sel, h->d_sel_count, h->d_status, -1);
        SGX_LAUNCH(kc, ogrid, dim3(oct_threads), stream, g, h->d_cand, h->d_cand_count, h->d_node_scratch, h->d_sel, h->d_sel_count, h->d_status, SGX_CAND_LDS);
    }
}

extern "C" int sgx_orb_extract_batch_dev(sgx_orb *h, const uint8_t *d_gray, int pitch, int batch,
                                         sgx_keypoint *d_kps, uint8_t *d_desc, int32_t *d_count, int cap, void *stream_)
{
    if (!h || !d_gray || !d_kps || !d_desc || !d_count) return SGX_ERR_INVALID;
    if (batch < 1 || batch > h->cfg.max_batch || pitch < h->g.W || (pitch & 3) || ((uintptr_t)d_gray & 3) || cap < h->g.kp_cap) return SGX_ERR_INVALID;
    sgx_stream_t stream = (sgx_stream_t)stream_;
    const SgxOrbGeom &g = h->g;
    const int nl = g.nlevels;
    h->last_batch = batch;
    SGX_CHECK_HIP(hipMemsetAsync(h->d_cand_count, 0, (size_t)batch * nl * 4, stream));
    sgx_prof_begin(SGX_K_RESIZE, stream);
    static const int pyr_threads = sgx_getenv("SGX_TUNE_PYR_THREADS") ? atoi(sgx_getenv("SGX_TUNE_PYR_THREADS")) : 512;      // workgroup size (measured: 0.130 / 0.086 / 0.073 ms per 64 frames at 128 / 256 / 512); env = tuning tap (64..1024)
    if (h->pyr_tiles > 0 && !g_orb_unfused_pyramid)
        SGX_LAUNCH_DYN(k_pyramid, dim3(h->pyr_tiles, batch), dim3(pyr_threads), h->pyr_lds, stream, g, h->pyr_tabs, d_gray, pitch, h->d_pyr, h->d_xt_all, h->d_yt_all, h->d_pyr_rects);
    else
        for (int l = 1; l < nl; l++) {
            dim3 grid((g.lv[l].w + 255) / 256, (g.lv[l].h + 3) / 4, batch);
            SGX_LAUNCH(k_resize, grid, dim3(256), stream, g, l, d_gray, pitch, h->d_pyr, h->d_xt[l], h->d_yt[l]);
        }
    sgx_prof_end(SGX_K_RESIZE, stream);
    sgx_prof_begin(SGX_K_FAST, stream);
    static const int fast_threads = sgx_getenv("SGX_TUNE_FAST_THREADS") ? atoi(sgx_getenv("SGX_TUNE_FAST_THREADS")) : 128;      // workgroup size; env = tuning tap (64..SGX_FAST_THREADS)
    static const int e_extra = sgx_getenv("SGX_TUNE_E_EXTRA_LDS") ? atoi(sgx_getenv("SGX_TUNE_E_EXTRA_LDS")) : 0;   // tuning tap: pad the extraction kernels' LDS to cap their occupancy
    SGX_LAUNCH_DYN(k_fast_cells, dim3(g.ncells * batch), dim3(fast_threads), g.fast_lds_bytes + e_extra, stream, g, h->d_cells, d_gray, pitch, h->d_pyr, batch,
               h->d_cand, h->d_cand_count, h->d_status);
    sgx_prof_end(SGX_K_FAST, stream);
    sgx_prof_begin(SGX_K_OCTREE, stream);
    launch_octree(h, batch, stream);
    sgx_prof_end(SGX_K_OCTREE, stream);
    sgx_prof_begin(SGX_K_ORIENT_DESC, stream);
    unsigned long long umax_packed = 0;
    for (int i = 0; i < 16; i++) umax_packed |= (unsigned long long)(h->umax_h[i] & 15) << (4 * i);
    // default: blur whole levels once (k_blur_levels), then a light per-keypoint kernel; SGX_TUNE_ORB_PATCH_BLUR=1 selects the first design (blur of a
    // 37x37 window per keypoint inside k_orient_desc) — identical bytes (tests), ~45 vs ~20+ VALU operations per pixel-equivalent
#ifdef SGX_DEBUG_TAPS      // the superseded descriptor kernels (k_orient_desc: blur per keypoint window; k_orient_desc2: one keypoint per wave) exist in the tap build only
    static const bool patch_blur = sgx_getenv("SGX_TUNE_ORB_PATCH_BLUR") != nullptr;
    if (patch_blur) {
        SGX_LAUNCH_DYN(k_orient_desc, dim3(g.kp_cap * batch), dim3(64), e_extra / 2, stream, g, d_gray, pitch, h->d_pyr, h->d_sel, h->d_sel_count,
                       umax_packed, h->d_pattern, (uint8_t *)d_kps, d_desc, d_count, cap, batch, h->d_status);
    } else
#endif
    {
        static const int blur_threads = sgx_getenv("SGX_TUNE_BLUR_THREADS") ? atoi(sgx_getenv("SGX_TUNE_BLUR_THREADS")) : 256;   // env = tuning tap (64..512)
        {   // persistent workgroups: `parts` per frame, each walks a contiguous range of the frame's tiles (see k_blur_levels); about 4 096 workgroups = 16 per CU
            const int blur_grid = sgx_getenv("SGX_TUNE_ORB_BLUR_GRID") ? atoi(sgx_getenv("SGX_TUNE_ORB_BLUR_GRID")) : 4096;          // tuning tap: target grid size (measured at 512 frames with the round-3 walk: 0.59 / 0.48 / 0.46 ms at 1 024 / 2 048 / 4 096)
            const int parts = std::min(g.nblur_tiles, std::max(1, blur_grid / batch));
            SGX_LAUNCH(k_blur_levels, dim3(parts * batch), dim3(std::min(512, std::max(256, blur_threads))), stream, g, h->d_blur_tiles, d_gray, pitch, h->d_pyr, h->d_blur, batch);
        }
#ifdef SGX_DEBUG_TAPS
        static const bool one_per_wave = sgx_getenv("SGX_TUNE_ORB_DESC_ONE_PER_WAVE") != nullptr;       // tuning tap: k_orient_desc2 (one keypoint per wave)
        if (one_per_wave)
            SGX_LAUNCH(k_orient_desc2, dim3(g.kp_cap * batch), dim3(64), stream, g, d_gray, pitch, h->d_pyr, h->d_blur, h->d_sel, h->d_sel_count,
                       umax_packed, h->d_pattern, (uint8_t *)d_kps, d_desc, d_count, cap, batch, h->d_status);
        else
#endif
            SGX_LAUNCH(k_orient_desc4, dim3(((g.kp_cap + 3) / 4) * batch), dim3(64), stream, g, d_gray, pitch, h->d_pyr, h->d_blur, h->d_sel, h->d_sel_count,
                       umax_packed, h->d_pattern, (uint8_t *)d_kps, d_desc, d_count, cap, batch, h->d_status);
    }
    sgx_prof_end(SGX_K_ORIENT_DESC, stream);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
}

extern "C" int sgx_orb_last_status(sgx_orb *h, void *stream_)
{
    if (!h) return SGX_ERR_INVALID;
    uint32_t st = 0;
    SGX_CHECK_HIP(hipMemcpyAsync(&st, h->d_status, 4, hipMemcpyDeviceToHost, (sgx_stream_t)stream_));
    SGX_CHECK_HIP(hipStreamSynchronize((sgx_stream_t)stream_));
    if (st) { SGX_CHECK_HIP(hipMemsetAsync(h->d_status, 0, 4, (sgx_stream_t)stream_)); return SGX_ERR_OVERFLOW; }
    return SGX_OK;
}

extern "C" int sgx_orb_extract(sgx_orb *h, const uint8_t *gray, int stride, sgx_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    if (!h || !gray || !kps || !desc || !n || stride < h->g.W || cap < 0) return SGX_ERR_INVALID;
    const int W = h->g.W, H = h->g.H, kc = h->g.kp_cap, P = (W + 3) & ~3;      // any image width (KITTI: 1241): the device copy is pitched to dwords
    if (stride == P) SGX_CHECK_HIP(hipMemcpyAsync(h->d_gray1, gray, (size_t)P * (H - 1) + W, hipMemcpyHostToDevice, 0));
    else for (int y = 0; y < H; y++)
        SGX_CHECK_HIP(hipMemcpyAsync(h->d_gray1 + (size_t)y * P, gray + (size_t)y * stride, W, hipMemcpyHostToDevice, 0));
    int rc = sgx_orb_extract_batch_dev(h, h->d_gray1, P, 1, (sgx_keypoint *)h->d_kps1, h->d_desc1, h->d_count1, kc, 0);
    if (rc != SGX_OK) return rc;
    int cnt = 0;
    SGX_CHECK_HIP(hipMemcpyAsync(&cnt, h->d_count1, 4, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    rc = sgx_orb_last_status(h, 0);
    if (rc != SGX_OK) return rc;
    if (cnt > cap) return SGX_ERR_OVERFLOW;
    SGX_CHECK_HIP(hipMemcpyAsync(kps, h->d_kps1, (size_t)cnt * 28, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipMemcpyAsync(desc, h->d_desc1, (size_t)cnt * 32, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    *n = cnt;
    return SGX_OK;
}

SGX_TAP int sgx_orb_debug_level_geometry(const sgx_orb *h, int level, int32_t *w, int32_t *hgt, int32_t *stride)
{
    if (!h || level < 0 || level >= h->g.nlevels) return SGX_ERR_INVALID;
    if (w) *w = h->g.lv[level].w; if (hgt) *hgt = h->g.lv[level].h; if (stride) *stride = h->g.lv[level].stride;
    return SGX_OK;
}

SGX_TAP int sgx_orb_debug_read_level(sgx_orb *h, int frame, int level, uint8_t *dst)
{
    if (!h || !dst || level < 1 || level >= h->g.nlevels || frame < 0 || frame >= h->cfg.max_batch) return SGX_ERR_INVALID;
    const SgxLevel &L = h->g.lv[level];
    for (int y = 0; y < L.h; y++)
        SGX_CHECK_HIP(hipMemcpyAsync(dst + (size_t)y * L.w, h->d_pyr + (size_t)frame * h->g.pyr_pitch + L.off + (size_t)y * L.stride, L.w, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    return SGX_OK;
}

SGX_TAP int sgx_orb_debug_read_candidates(sgx_orb *h, int frame, int level, int32_t *x, int32_t *y, int32_t *score, int cap, int *n)
{
    if (!h || !n || level < 0 || level >= h->g.nlevels || frame < 0 || frame >= h->cfg.max_batch) return SGX_ERR_INVALID;
    int cnt = 0;
    SGX_CHECK_HIP(hipMemcpyAsync(&cnt, h->d_cand_count + frame * h->g.nlevels + level, 4, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    if (cnt > h->g.lv[level].cand_cap) cnt = h->g.lv[level].cand_cap;
    std::vector<uint32_t> buf(cnt > 0 ? cnt : 1);
    SGX_CHECK_HIP(hipMemcpyAsync(buf.data(), h->d_cand + (size_t)frame * h->g.cand_pitch + h->g.lv[level].cand_off, (size_t)cnt * 4, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    for (int i = 0; i < cnt && i < cap; i++) { x[i] = buf[i] & 0xFFF; y[i] = (buf[i] >> 12) & 0xFFF; score[i] = buf[i] >> 24; }
    *n = cnt;
    return SGX_OK;
}

// test tap: run k_octree alone on a caller-supplied candidate list for `level` (frame slot 0)
SGX_TAP int sgx_orb_debug_run_octree(sgx_orb *h, int level, const uint32_t *packed, int n, uint32_t *out_sel, int cap, int *nsel)
{
    if (!h || !packed || !out_sel || !nsel || level < 0 || level >= h->g.nlevels || n < 0 || n > h->g.lv[level].cand_cap) return SGX_ERR_INVALID;
    const int nl = h->g.nlevels;
    std::vector<int> cnt(nl, 0); cnt[level] = n;
    SGX_CHECK_HIP(hipMemcpyAsync(h->d_cand_count, cnt.data(), nl * 4, hipMemcpyHostToDevice, 0));
    SGX_CHECK_HIP(hipMemcpyAsync(h->d_cand + h->g.lv[level].cand_off, packed, (size_t)n * 4, hipMemcpyHostToDevice, 0));
    launch_octree(h, 1, (sgx_stream_t)0);
    int ns = 0;
    SGX_CHECK_HIP(hipMemcpyAsync(&ns, h->d_sel_count + level, 4, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    if (ns > cap) return SGX_ERR_OVERFLOW;
    SGX_CHECK_HIP(hipMemcpyAsync(out_sel, h->d_sel + (size_t)level * SGX_OCT_MAXN, (size_t)ns * 4, hipMemcpyDeviceToHost, 0));
    SGX_CHECK_HIP(hipStreamSynchronize(0));
    *nsel = ns;
    return sgx_orb_last_status(h, 0);
}
