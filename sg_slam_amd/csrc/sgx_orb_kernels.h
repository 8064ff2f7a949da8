// sgx_orb_kernels.h — HIP kernels of the ORB extractor (gfx950), phase-style (see sgx_rt.h).
//
// Pipeline for a batch of B frames (each kernel covers all B frames in one launch):
//   k_resize      x (nlevels-1)  level l <- level l-1     (cv::resize INTER_LINEAR, fixed point)
//   k_fast_cells  x 1            per-cell FAST-9/16 score + NMS + 20/7 threshold fallback
//   k_octree      x 1            DistributeOctTree, one workgroup per (frame, level)
//   k_orient_desc x 1            IC_Angle + on-the-fly 7x7 Gaussian + steered BRIEF, one wave per keypoint
// Reference behaviour: src/sg-slam/src/ORBextractor.cc (cited per kernel).  All arithmetic is
// integer or fp32 evaluated exactly as written (-ffp-contract=off; IEEE div/sqrt).
#pragma once
#include "sgx_rt.h"
#include "sgx_block.h"

#define SGX_MAX_LEVELS 12
#define SGX_EDGE 19            /* EDGE_THRESHOLD, ORBextractor.cc:75 */
#define SGX_BORDER 16          /* EDGE_THRESHOLD-3 = minBorderX/Y, ORBextractor.cc:774 */
#define SGX_TILE_MAX 68        /* max FAST cell tile edge incl. 3-px aprons */
#define SGX_TILE_STRIDE 72
#define SGX_CAND_LDS 8192      /* k_octree keeps up to this many keys of a (frame, level) in LDS; more -> global-memory variant */
#define SGX_OCT_MAXN 1280      /* max octree list length (per-level quota + 3) */
#define SGX_OCT_THREADS 512      /* measured: 0.0435 ms per 64 frames at 512 threads vs 0.0475 / 0.056 / 0.071 at 256 / 128 / 64 */

struct SgxLevel {
    int w, h, stride;          // level image geometry (stride in bytes)
    int off;                   // byte offset of the level inside one frame's pyramid buffer (levels >= 1)
    int quota;                 // mnFeaturesPerLevel[level]
    int ncols, nrows, wcell, hcell;  // FAST cell grid (ORBextractor.cc:785-788)
    int cell0;                 // index of this level's first cell in the cell table
    int patch_size;            // (int)(31*scale)
    int cand_off, cand_cap;    // slice of one frame's candidate buffer (cap = exact upper bound of NMS survivors)
    int boff, bstride;         // blurred copy of the level (all levels incl. 0): byte offset inside one frame's blur buffer, row stride
    float scale;               // mvScaleFactor[level]
};

struct SgxOrbGeom {
    int nlevels, W, H;
    int pyr_pitch;             // bytes per frame of pyramid storage (levels 1..)
    int ncells;                // total valid cells over all levels
    int kp_cap;                // keypoint capacity per frame
    int fast_off_score, fast_off_qlist, fast_off_out, fast_lds_bytes;   // k_fast_cells dynamic-LDS carve (bytes)
    int cand_pitch;            // candidate entries per frame (all levels)
    int blur_pitch, nblur_tiles;   // bytes per frame of blurred-level storage; tiles of k_blur_levels over all levels
    int ini_th, min_th;
    SgxLevel lv[SGX_MAX_LEVELS];
};

struct SgxCell { short level, x0, y0, cw, ch, ox, oy, pad; };  // tile rect in level coords; ox=j*wCell, oy=i*hCell; pad = ceil(2^16 / ng), ng = dword groups per tile row (k_fast_cells)

// status bits written by kernels
#define SGX_ST_CAND_OVERFLOW 1u
#define SGX_ST_NODE_OVERFLOW 2u
#define SGX_ST_KP_OVERFLOW 4u

SGX_DEV const uint8_t *sgx_level_ptr(const SgxOrbGeom &g, const uint8_t *gray, int gray_pitch, const uint8_t *pyr,
                                     int frame, int level, int *stride)
{
    if (level == 0) { *stride = gray_pitch; return gray + (size_t)frame * gray_pitch * g.H; }
    *stride = g.lv[level].stride;
    return pyr + (size_t)frame * g.pyr_pitch + g.lv[level].off;
}

// ---------------------------------------------------------------------------------------------
// k_resize: cv::resize(u8, INTER_LINEAR) level l-1 -> l (ORBextractor.cc:1121).  The per-column
// {sx, a0, a1} and per-row {sy0, sy1, b0, b1} tables are built on the host with the exact
// OpenCV float/cvRound sequence (sgx_orb.cpp: build_resize_tables), so the kernel is integer-only:
//   h0 = S0[sx]*a0 + S0[sx+1]*a1 ; h1 likewise ; dst = (((b0*(h0>>4))>>16) + ((b1*(h1>>4))>>16) + 2) >> 2
// Each thread produces 4 adjacent pixels and stores one dword.  grid = (ceil(dw/4/64), ceil(dh/4), B).
// ---------------------------------------------------------------------------------------------
struct SgxXTab { short sx, sx1, a0, a1; };
struct SgxYTab { short sy0, sy1, b0, b1; };

SGX_KERNEL(256) k_resize(SgxOrbGeom g, int level, const uint8_t *gray, int gray_pitch, uint8_t *pyr,
                         const SgxXTab *xt, const SgxYTab *yt)
{
    SGX_THREADS_BEGIN(tid)
    const int dw = g.lv[level].w, dh = g.lv[level].h;
    const int x4 = ((int)blockIdx.x * 64 + (tid & 63)) * 4;
    const int y = (int)blockIdx.y * 4 + (tid >> 6);
    const int frame = (int)blockIdx.z;
    if (x4 < dw && y < dh) {
        int sstride, dstride;
        const uint8_t *src = sgx_level_ptr(g, gray, gray_pitch, pyr, frame, level - 1, &sstride);
        uint8_t *dst = (uint8_t *)sgx_level_ptr(g, gray, gray_pitch, pyr, frame, level, &dstride);
        const SgxYTab ty = yt[y];
        const uint8_t *r0 = src + (size_t)ty.sy0 * sstride, *r1 = src + (size_t)ty.sy1 * sstride;
        uint32_t out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = x4 + i;
            if (x < dw) {
                const SgxXTab tx = xt[x];
                const int h0 = r0[tx.sx] * tx.a0 + r0[tx.sx1] * tx.a1;
                const int h1 = r1[tx.sx] * tx.a0 + r1[tx.sx1] * tx.a1;
                const int v = (((ty.b0 * (h0 >> 4)) >> 16) + ((ty.b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                out |= (uint32_t)(v & 255) << (8 * i);
            }
        }
        *(uint32_t *)(dst + (size_t)y * dstride + x4) = out;   // stride is a multiple of 64 -> in-bounds, aligned
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_pyramid: ORBextractor::ComputePyramid (ORBextractor.cc:1108-1133) for all levels in ONE launch.  The chained resizes
// level l-1 -> l are evaluated tile by tile: a workgroup owns one tile of every level (a fixed 1/(ntx*nty) partition of each
// level, x boundaries multiples of 4), stages the level-0 footprint of its tiles in LDS, and walks down the chain with two
// ping-pong LDS buffers, computing at each level the region the next level needs plus the part it owns (a few halo columns /
// rows are recomputed by neighbouring workgroups), and writing only the owned part to HBM as dwords.  Same tables and integer
// formula as k_resize, so the pyramid is byte-identical; level l-1 is never re-read from HBM.
// Host-built per (tile, level) rects: needed region (x0 and width multiples of 4) and owned region.  grid = (ntiles, B).
// ---------------------------------------------------------------------------------------------
struct SgxPyrRect { short nx0, ny0, nw, nh, ox0, oy0, ox1, oy1; unsigned qmagic; int xo, yo, pad; };   // qmagic = ceil(2^32 / (nw/4)); xo / yo: slots of this level's table slices in LDS
struct SgxPyrTabs { int xoff[SGX_MAX_LEVELS], yoff[SGX_MAX_LEVELS]; int lds_a, lds_b, lds_x; };        // LDS carve: [buffer A | buffer B | x-table slices | y-table slices]

SGX_DEV unsigned sgx_udiv_magic(unsigned n, unsigned m)
{
#ifndef SGX_EMU
    return m ? __umulhi(n, m) : n;
#else
    return m ? (unsigned)(((unsigned long long)n * m) >> 32) : n;
#endif
}

SGX_KERNEL(1024) k_pyramid(SgxOrbGeom g, SgxPyrTabs tb, const uint8_t *gray, int gray_pitch, uint8_t *pyr, const SgxXTab *xt, const SgxYTab *yt,
                          const SgxPyrRect *rects)
{
    SGX_DYN_LDS(smem);
    const int tile = (int)blockIdx.x, frame = (int)blockIdx.y, nl = g.nlevels;
    const SgxPyrRect *R = rects + (size_t)tile * nl;
    uint8_t *cur = smem, *nxt = smem + tb.lds_a;
    SgxXTab *xtl_lds = (SgxXTab *)(smem + tb.lds_a + tb.lds_b); SgxYTab *ytl_lds = (SgxYTab *)(smem + tb.lds_a + tb.lds_b + tb.lds_x);
    {   // level-0 footprint, dword loads (x0 and width are multiples of 4; the row pitch is a multiple of 4) + this tile's slices of every level's tables
        const SgxPyrRect r0 = R[0];
        const uint8_t *src = gray + (size_t)frame * gray_pitch * g.H;
        const int q = r0.nw >> 2, groups = q * r0.nh;
        SGX_THREADS_BEGIN(tid)
        for (int idx = tid; idx < groups; idx += (int)blockDim.x) {
            const int y = (int)sgx_udiv_magic((unsigned)idx, r0.qmagic), x4 = (idx - y * q) * 4;
            *(uint32_t *)(cur + y * r0.nw + x4) = *(const uint32_t *)(src + (size_t)(r0.ny0 + y) * gray_pitch + r0.nx0 + x4);
        }
        for (int l = 1; l < nl; l++) {
            const SgxPyrRect r = R[l];
            const int nx = min((int)r.nw, g.lv[l].w - r.nx0);
            for (int i = tid; i < nx; i += (int)blockDim.x) xtl_lds[r.xo + i] = xt[tb.xoff[l] + r.nx0 + i];
            for (int i = tid; i < r.nh; i += (int)blockDim.x) ytl_lds[r.yo + i] = yt[tb.yoff[l] + r.ny0 + i];
        }
        SGX_THREADS_END
    }
    SGX_SYNC();
    for (int l = 1; l < nl; l++) {
        const SgxPyrRect rp = R[l - 1], r = R[l];
        const int W = g.lv[l].w, dstride = g.lv[l].stride;
        uint8_t *dst = pyr + (size_t)frame * g.pyr_pitch + g.lv[l].off;
        const SgxXTab *xtl = xtl_lds + r.xo - r.nx0; const SgxYTab *ytl = ytl_lds + r.yo - r.ny0;
        const int q = r.nw >> 2, groups = q * r.nh;
        SGX_THREADS_BEGIN(tid)
        for (int idx = tid; idx < groups; idx += (int)blockDim.x) {
            const int y = (int)sgx_udiv_magic((unsigned)idx, r.qmagic), x4 = (idx - y * q) * 4;
            const int gy = r.ny0 + y, gx4 = r.nx0 + x4;
            const SgxYTab ty = ytl[gy];
            const uint8_t *r0 = cur + (ty.sy0 - rp.ny0) * rp.nw - rp.nx0, *r1 = cur + (ty.sy1 - rp.ny0) * rp.nw - rp.nx0;
            uint32_t out = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x = gx4 + i;
                if (x < W) {
                    const SgxXTab tx = xtl[x];
                    const int h0 = r0[tx.sx] * tx.a0 + r0[tx.sx1] * tx.a1;
                    const int h1 = r1[tx.sx] * tx.a0 + r1[tx.sx1] * tx.a1;
                    const int v = (((ty.b0 * (h0 >> 4)) >> 16) + ((ty.b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    out |= (uint32_t)(v & 255) << (8 * i);
                }
            }
            *(uint32_t *)(nxt + y * r.nw + x4) = out;
            if (gy >= r.oy0 && gy < r.oy1 && gx4 >= r.ox0 && gx4 < r.ox1) *(uint32_t *)(dst + (size_t)gy * dstride + gx4) = out;   // stride is a multiple of 64: in-bounds, aligned
        }
        SGX_THREADS_END
        SGX_SYNC();
        uint8_t *t = cur; cur = nxt; nxt = t;
    }
}

// ---------------------------------------------------------------------------------------------
// k_fast_cells: ORBextractor.cc:790-830 for every cell of every level of every frame.
// One 256-thread workgroup per (cell, frame).  cv::FAST(cell, 20, nms) with fallback
// cv::FAST(cell, 7, nms) is evaluated from ONE threshold-free score map S = A-1, where
// A = max over the 16 nine-pixel arcs of min(v - ring) (bright) / min(ring - v) (dark):
// a pixel is a corner at threshold t iff S >= t, its OpenCV response is S, NMS is a strict
// '>' against the 8 neighbours inside the cell interior (everything else scores 0), and the
// cell keeps {NMS-max, S>=20} unless that set is empty, then {NMS-max, S>=7}.
// Candidates are appended unordered to the (frame, level) list as packed x | y<<12 | S<<24
// with x,y relative to the (16,16) border origin (ORBextractor.cc:823-824); k_octree does not
// depend on their order.
// ---------------------------------------------------------------------------------------------
// ((hi:lo) >> 8*sh) as 32 bits (v_alignbyte_b32)
SGX_DEV uint32_t sgx_alignbyte(uint32_t hi, uint32_t lo, int sh)
{
#ifndef SGX_EMU
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)sh);
#else
    return (uint32_t)((((unsigned long long)hi << 32) | lo) >> (8 * sh));
#endif
}

// exact integer dot products of packed operands: 4 x u8 (v_dot4_u32_u8) and 2 x u16 (v_dot2_u32_u16), 32-bit accumulate, no clamp
SGX_DEV uint32_t sgx_udot4(uint32_t a, uint32_t b, uint32_t c)
{
#ifndef SGX_EMU
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
    return c;
#endif
}
SGX_DEV uint32_t sgx_udot2(uint32_t a, uint32_t b, uint32_t c)
{
#ifndef SGX_EMU
    typedef unsigned short sgx_us2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_udot2(__builtin_bit_cast(sgx_us2, a), __builtin_bit_cast(sgx_us2, b), c, false);
#else
    return c + (a & 0xFFFFu) * (b & 0xFFFFu) + (a >> 16) * (b >> 16);
#endif
}

// packed 2 x u16 helpers (v_pk_sub_u16 clamp / v_pk_min_u16 on the device)
#ifndef SGX_EMU
typedef unsigned short sgx_u16x2 __attribute__((ext_vector_type(2)));
SGX_DEV uint32_t sgx_pk_usubsat_u16(uint32_t a, uint32_t b)
{
    const sgx_u16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(sgx_u16x2, a), __builtin_bit_cast(sgx_u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
SGX_DEV uint32_t sgx_pk_min_u16(uint32_t a, uint32_t b)
{
    const sgx_u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(sgx_u16x2, a), __builtin_bit_cast(sgx_u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
SGX_DEV uint32_t sgx_pk_max_u16(uint32_t a, uint32_t b)
{
    const sgx_u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(sgx_u16x2, a), __builtin_bit_cast(sgx_u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
}
// bytes (0, 2) or (1, 3) of a dword as 2 x u16: one v_perm_b32 (selector byte 0x0c = constant zero) instead of shift + mask
SGX_DEV uint32_t sgx_even_bytes_u16(uint32_t v) { return __builtin_amdgcn_perm(0u, v, 0x0c020c00u); }
SGX_DEV uint32_t sgx_odd_bytes_u16(uint32_t v) { return __builtin_amdgcn_perm(0u, v, 0x0c030c01u); }
#else
SGX_DEV uint32_t sgx_pk_usubsat_u16(uint32_t a, uint32_t b)
{
    const uint32_t al = a & 0xFFFFu, ah = a >> 16, bl = b & 0xFFFFu, bh = b >> 16;
    return (al > bl ? al - bl : 0u) | ((ah > bh ? ah - bh : 0u) << 16);
}
SGX_DEV uint32_t sgx_pk_min_u16(uint32_t a, uint32_t b)
{
    const uint32_t al = a & 0xFFFFu, ah = a >> 16, bl = b & 0xFFFFu, bh = b >> 16;
    return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16);
}
SGX_DEV uint32_t sgx_pk_max_u16(uint32_t a, uint32_t b)
{
    const uint32_t al = a & 0xFFFFu, ah = a >> 16, bl = b & 0xFFFFu, bh = b >> 16;
    return (al > bl ? al : bl) | ((ah > bh ? ah : bh) << 16);
}
SGX_DEV uint32_t sgx_even_bytes_u16(uint32_t v) { return v & 0x00FF00FFu; }
SGX_DEV uint32_t sgx_odd_bytes_u16(uint32_t v) { return (v >> 8) & 0x00FF00FFu; }
#endif

// ((hi:lo) >> sh) as 32 bits (v_alignbit_b32); with sh = 31 it shifts the sign bit of lo into hi from the right
SGX_DEV uint32_t sgx_alignbit(uint32_t hi, uint32_t lo, int sh)
{
#ifndef SGX_EMU
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
#else
    return (uint32_t)((((unsigned long long)hi << 32) | lo) >> sh);
#endif
}

SGX_DEV uint32_t sgx_has9(uint32_t m16)
{
    uint32_t m = m16 | (m16 << 16);
    uint32_t a = m & (m >> 1);
    uint32_t b = a & (a >> 2);
    uint32_t c = b & (b >> 4);
    return (c & (m >> 8)) & 0xFFFFu;
}

#define SGX_FAST_THREADS 256     /* launch bound; the host launches 128 (2 waves per cell: measured best on MI355X — 0.184 ms per 64 frames vs 0.202 / 0.216 / 0.287 at 192 / 256 / 320) */
SGX_KERNEL(SGX_FAST_THREADS) k_fast_cells(SgxOrbGeom g, const SgxCell *cells, const uint8_t *gray, int gray_pitch, const uint8_t *pyr,
                             int batch, uint32_t *cand, int *cand_count, uint32_t *status)
{
    // LDS: carved from one dynamic pool sized on the host from the largest tile of this geometry (g.fast_* fields), so that
    // the common 640x480 case (tiles <= 43x40) runs 8 workgroups per CU instead of the 5 a worst-case static layout allows.
    SGX_DYN_LDS(pool);
    uint32_t *tile_dw = (uint32_t *)pool;                                   // staged rows, 72 B stride; tile column x at byte lead + x
    uint8_t *score = (uint8_t *)pool + g.fast_off_score;                    // threshold-free FAST score map, same layout
    uint16_t *qlist = (uint16_t *)((uint8_t *)pool + g.fast_off_qlist);     // survivors of the quick test; bit 15 = is a corner
    uint32_t *outbuf = (uint32_t *)((uint8_t *)pool + g.fast_off_out);      // NMS survivors (packed candidates)
    SGX_LDS int n_lo, out_base, n_quick;
    const uint8_t *tile = (const uint8_t *)tile_dw;

    // block -> (cell, frame): frame fastest so that frame f stays on XCD f%8 (block b -> XCD b%8)
    const int bid = (int)blockIdx.x;
    const int frame = bid % batch, cid = bid / batch;
    const SgxCell c = cells[cid];
    const int cw = c.cw, ch = c.ch, level = c.level;
    int stride;
    const uint8_t *img = sgx_level_ptr(g, gray, gray_pitch, pyr, frame, level, &stride);
    const int thr_lo = g.min_th, thr_hi = g.ini_th;
    const int xa = c.x0 & ~3, lead = c.x0 - xa, ndw = (lead + cw + 3) >> 2;
    const int SD = SGX_TILE_STRIDE / 4;

    // phase A: stage the tile rows as aligned dwords (coalesced global loads, one LDS dword store each); zero the score map
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { n_lo = 0; n_quick = 0; }
    for (int i = tid; i < ch * SD; i += (int)blockDim.x) {
        const int r = i / SD, q = i - r * SD;
        tile_dw[i] = q < ndw ? *(const uint32_t *)(img + (size_t)(c.y0 + r) * stride + xa + 4 * q) : 0u;
    }
    for (int i = tid; i < ch * SD; i += (int)blockDim.x) ((uint32_t *)score)[i] = 0u;       // score map (72-byte rows) as dwords
    SGX_THREADS_END
    SGX_SYNC();

    // Two passes, as the reference: FAST at iniThFAST first; only a cell that yields no corner is searched again at minThFAST (ORBextractor.cc:806-817).  The
    // segment test, the threshold-free score and the 3x3 NMS of a pass see exactly the corners cv::FAST(threshold) sees: a pixel is a corner at threshold t iff its
    // score >= t, and a neighbour that is not a corner at t has a score below t, so it cannot suppress one that is (cv::FAST scores it 0).  Textured cells — the
    // common case — never pay for the low-threshold pass, whose quick test lets several times more pixels through.
    const int ih = ch - 6, ng = (lead + cw + 3) >> 2;
    const unsigned ng_m16 = (unsigned)(unsigned short)c.pad;
    for (int pass = 0; pass < 2; pass++) {
        const int thr = pass == 0 ? thr_hi : thr_lo;
        // phase B1: necessary condition on every interior pixel.  A 9-pixel arc of the 16-ring contains one pixel of every antipodal
        // pair, so a corner needs (p0|p8) & (p4|p12) all-brighter or all-darker (the high-speed test cv::FAST itself starts with).
        // A task = 4 horizontally adjacent pixels: the compass pixels come from 7 aligned LDS dwords.  Survivors are compacted.
        SGX_THREADS_BEGIN(tid)
        for (int t = tid; t < ih * ng; t += (int)blockDim.x) {
            const int yq = (int)(((unsigned)t * ng_m16) >> 16), y = 3 + yq, gq = t - yq * ng;          // t / ng, t % ng with the host's 16-bit reciprocal (exact for t < 4096, ng <= 16)
            const int x0 = 4 * gq + 3 - lead;                                   // tile column of the first pixel of the group
            if (x0 + 3 < 3 || x0 >= cw - 3) continue;
            const uint32_t *rc = tile_dw + y * SD + gq, *rp = rc + 3 * SD, *rm = rc - 3 * SD;
            const bool has1 = gq + 1 < SD, has2 = gq + 2 < SD;
            const uint32_t C0 = rc[0], C1 = has1 ? rc[1] : 0u, C2 = has2 ? rc[2] : 0u;
            const uint32_t P0 = rp[0], P1 = has1 ? rp[1] : 0u, M0 = rm[0], M1 = has1 ? rm[1] : 0u;
            // the 4 pixels of the task as packed bytes: centre v (tile bytes 3..6 of the group), the compass pixels below / above (same columns of rows
            // y+3 / y-3) and right / left (bytes 6..9 / 0..3), then as two sets of 2 x u16 (even / odd pixels) for the packed saturating compares:
            //   brighter  <=>  usubsat(r, v + t) != 0      darker  <=>  usubsat(usubsat(v, t), r) != 0
            const uint32_t V = sgx_alignbyte(C1, C0, 3), R4 = sgx_alignbyte(C2, C1, 2), R12 = C0, R0 = sgx_alignbyte(P1, P0, 3), R8 = sgx_alignbyte(M1, M0, 3);
            const uint32_t T2 = (uint32_t)thr * 0x00010001u;
            uint32_t any[2];
#if !defined(SGX_FAST_QUICK_V2)      /* the round-3 formulation; V2 (below) is the round-4 attempt kept for the record: 17 fewer instructions per task, 5 % SLOWER (A/B, tools/ab_run.sh) */
    #pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                const int sh = 8 * s2;
                const uint32_t v = (V >> sh) & 0x00FF00FFu, r0 = (R0 >> sh) & 0x00FF00FFu, r8 = (R8 >> sh) & 0x00FF00FFu, r4 = (R4 >> sh) & 0x00FF00FFu, r12 = (R12 >> sh) & 0x00FF00FFu;
                const uint32_t hi = v + T2, lo = sgx_pk_usubsat_u16(v, T2);
                const uint32_t br = sgx_pk_min_u16(sgx_pk_usubsat_u16(r0, hi) | sgx_pk_usubsat_u16(r8, hi), sgx_pk_usubsat_u16(r4, hi) | sgx_pk_usubsat_u16(r12, hi));
                const uint32_t dk = sgx_pk_min_u16(sgx_pk_usubsat_u16(lo, r0) | sgx_pk_usubsat_u16(lo, r8), sgx_pk_usubsat_u16(lo, r4) | sgx_pk_usubsat_u16(lo, r12));
                any[s2] = br | dk;
            }
#else
    #pragma unroll
            for (int s2 = 0; s2 < 2; s2++) {
                // round-4 attempt: (p0 | p8) & (p4 | p12) brighter  <=>  min(max(r0, r8), max(r4, r12)) > v + t, darker  <=>  max(min(r0, r8), min(r4, r12)) < v - t: two packed
                // max / min pairs and ONE saturating subtraction per polarity instead of four each; the 2 x u16 operands come out of the byte quads with one v_perm_b32.
                // Measured 1.158 against 1.106 ms per 512 frames: it trades 2-cycle logic / shift instructions for 4-cycle v_perm_b32 / v_pk_max_u16 (r2_ubench_valu_issue.txt)
                const uint32_t v = s2 ? sgx_odd_bytes_u16(V) : sgx_even_bytes_u16(V), r0 = s2 ? sgx_odd_bytes_u16(R0) : sgx_even_bytes_u16(R0), r8 = s2 ? sgx_odd_bytes_u16(R8) : sgx_even_bytes_u16(R8),
                               r4 = s2 ? sgx_odd_bytes_u16(R4) : sgx_even_bytes_u16(R4), r12 = s2 ? sgx_odd_bytes_u16(R12) : sgx_even_bytes_u16(R12);
                const uint32_t hi = v + T2, lo = sgx_pk_usubsat_u16(v, T2);
                const uint32_t br = sgx_pk_usubsat_u16(sgx_pk_min_u16(sgx_pk_max_u16(r0, r8), sgx_pk_max_u16(r4, r12)), hi);
                const uint32_t dk = sgx_pk_usubsat_u16(lo, sgx_pk_max_u16(sgx_pk_min_u16(r0, r8), sgx_pk_min_u16(r4, r12)));
                any[s2] = br | dk;
            }
#endif
            if ((any[0] | any[1]) == 0u) continue;                                   // three tasks in four hold no survivor
    #pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x = x0 + i;
                const uint32_t f = (any[i & 1] >> (16 * (i >> 1))) & 0xFFFFu;          // pixel i = half (i>>1) of set (i&1)
                if (x >= 3 && x < cw - 3 && f) {
                    const int slot = sgx_atomic_add(&n_quick, 1);
                    qlist[slot] = (uint16_t)(y * SGX_TILE_STRIDE + lead + x);
                }
            }
        }
        SGX_THREADS_END
        SGX_SYNC();

        // phase B2: full segment test (>= 9 contiguous ring pixels brighter than v+t or darker than v-t) on the survivors.
        // The two 16-bit ring masks are shifted in from sign bits with v_alignbit (2 VALU per ring pixel and polarity).
        SGX_THREADS_BEGIN(tid)
        for (int t = tid; t < n_quick; t += (int)blockDim.x) {
            const int pos = qlist[t];
            const uint8_t *p = tile + pos;
            const int v = p[0], lo = v - thr, hi = v + thr;
            int r[16];                                                    // the 16 ring pixels, clockwise from (0, +3) — loaded once for the mask test and the score
            r[0] = p[3 * SGX_TILE_STRIDE];       r[1] = p[3 * SGX_TILE_STRIDE + 1];  r[2] = p[2 * SGX_TILE_STRIDE + 2];
            r[3] = p[SGX_TILE_STRIDE + 3];       r[4] = p[3];                        r[5] = p[-SGX_TILE_STRIDE + 3];
            r[6] = p[-2 * SGX_TILE_STRIDE + 2];  r[7] = p[-3 * SGX_TILE_STRIDE + 1]; r[8] = p[-3 * SGX_TILE_STRIDE];
            r[9] = p[-3 * SGX_TILE_STRIDE - 1];  r[10] = p[-2 * SGX_TILE_STRIDE - 2]; r[11] = p[-SGX_TILE_STRIDE - 3];
            r[12] = p[-3];                       r[13] = p[SGX_TILE_STRIDE - 3];     r[14] = p[2 * SGX_TILE_STRIDE - 2];
            r[15] = p[3 * SGX_TILE_STRIDE - 1];
            uint32_t mb = 0, md = 0;
    #pragma unroll
            for (int k = 0; k < 16; k++) { mb = sgx_alignbit(mb, (uint32_t)(hi - r[k]), 31); md = sgx_alignbit(md, (uint32_t)(r[k] - lo), 31); }
            if (!(sgx_has9(mb & 0xFFFFu) | sgx_has9(md & 0xFFFFu))) continue;
            qlist[t] = (uint16_t)(pos | 0x8000);                          // pos < 68*72 < 2^15
            // threshold-free score of the corner, in the same pass (phase C of the first version cost one more barrier and one more walk of the list)
            int d[16];
    #pragma unroll
            for (int k = 0; k < 16; k++) d[k] = v - r[k];
            int mn2[16], mx2[16], mn4[16], mx4[16];
    #pragma unroll
            for (int k = 0; k < 16; k++) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
    #pragma unroll
            for (int k = 0; k < 16; k++) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
            int A = -512, Bm = 512;
    #pragma unroll
            for (int k = 0; k < 16; k++) {
                const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
                const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
                A = max(A, mn9); Bm = min(Bm, mx9);
            }
            const int s = max(A, -Bm) - 1;
            score[pos] = (uint8_t)s;
        }
        SGX_THREADS_END
        SGX_SYNC();

        // phase D: NMS (strict > over the 8 neighbours; apron and non-corners are 0), count survivors
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < n_quick; i += (int)blockDim.x) {
            if (!(qlist[i] & 0x8000)) continue;
            const int pos = qlist[i] & 0x7FFF;
            const uint8_t *s = score + pos;
            const int v = s[0];
            const bool mx = v > s[-1] && v > s[1] && v > s[-SGX_TILE_STRIDE - 1] && v > s[-SGX_TILE_STRIDE] && v > s[-SGX_TILE_STRIDE + 1] &&
                            v > s[SGX_TILE_STRIDE - 1] && v > s[SGX_TILE_STRIDE] && v > s[SGX_TILE_STRIDE + 1];
            if (mx && v >= thr) {
                const int slot = sgx_atomic_add(&n_lo, 1);
                const int y = pos / SGX_TILE_STRIDE, x = pos - y * SGX_TILE_STRIDE - lead;
                outbuf[slot] = (uint32_t)(x + c.ox) | ((uint32_t)(y + c.oy) << 12) | ((uint32_t)v << 24);
            }
        }
        SGX_THREADS_END
        SGX_SYNC();

        if (n_lo > 0 || thr_hi == thr_lo) break;                      // uniform: n_lo is final after the barrier above
        SGX_THREADS_BEGIN(tid) if (tid == 0) n_quick = 0; SGX_THREADS_END
        SGX_SYNC();
    }

    // phase E: reserve space in the (frame, level) list, emit the NMS survivors of the pass that found corners
    const int n_emit = n_lo;
    SGX_THREADS_BEGIN(tid)
    if (tid == 0 && n_emit > 0) out_base = sgx_atomic_add(&cand_count[frame * g.nlevels + level], n_emit);
    SGX_THREADS_END
    SGX_SYNC();
    if (n_emit > 0) {
        uint32_t *dst = cand + (size_t)frame * g.cand_pitch + g.lv[level].cand_off;
        const int dcap = g.lv[level].cand_cap;
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < n_emit; i += (int)blockDim.x) {
            const int slot = out_base + i;
            if (slot < dcap) dst[slot] = outbuf[i];
            else sgx_atomic_or(status, SGX_ST_CAND_OVERFLOW);
        }
        SGX_THREADS_END
    }
}

// ---------------------------------------------------------------------------------------------
// k_octree: ORBextractor::DistributeOctTree (ORBextractor.cc:540-764) + the coordinate shift,
// octave and size assignment of ComputeKeyPointsOctTree (:838-848).  One workgroup per
// (frame, level); candidates and the node list live in LDS.
//
// Data-parallel restatement of the reference's std::list algorithm.  The list is an array in
// list order; node id == list position.  One "pass" splits a set of nodes at once:
//   phase 1 (reference :607-666): every node with >1 key splits; children are push_front'ed in
//     creation order, so the new list is reverse(children in creation order) ++ kept nodes.
//   phase 2 (reference :677-738): the children created by the previous pass that hold >1 key are
//     processed in descending (size, creation order) [the reference sorts (size, node address),
//     :685; address ties are allocator dependent there — see oracle/orb_oracle.c], stopping at the
//     first split that makes the list reach N; same reverse-prepend rule for the list.
// Keys never move: key k carries node_of[k]; a split recomputes the quadrant of each key of a
// splitting node (DivideNode :482-538) and remaps node_of through the new positions.
// The final pick per node is the max response, first-in-candidate-order on ties (:745-761);
// candidate order in the reference is cell-raster, row-major inside a cell, which is encoded
// in a rank key so the unordered candidate list of k_fast_cells gives the same pick.
// ---------------------------------------------------------------------------------------------

struct SgxOctNode { uint16_t ulx, uly, urx, bry; };   // UL.x, UL.y, UR.x, BR.y — all DivideNode needs

// quadrant of key e inside node b (DivideNode :484-527): n1=0 (left/top) n2=1 (right/top) n3=2 n4=3
SGX_DEV int sgx_oct_quadrant(const SgxOctNode b, uint32_t e)
{
    const int halfX = (int)ceilf((float)(b.urx - b.ulx) / 2), halfY = (int)ceilf((float)(b.bry - b.uly) / 2);
    const float x = (float)(e & 0xFFF), y = (float)((e >> 12) & 0xFFF);
    const float sx = (float)(b.ulx + halfX), sy = (float)(b.uly + halfY);
    return (x < sx) ? ((y < sy) ? 0 : 2) : ((y < sy) ? 1 : 3);
}

// (response desc, candidate order asc) as one sortable 48-bit key; candidate order in the
// reference is cell-raster, row-major inside a cell (ORBextractor.cc:790-827)
SGX_DEV unsigned long long sgx_oct_pick_key(uint32_t e, int wcell, int hcell)
{
    const int x = (int)(e & 0xFFF), y = (int)((e >> 12) & 0xFFF);
    const int cj = (x - 3) / wcell, ci = (y - 3) / hcell;          // owning FAST cell (interiors tile the level)
    const unsigned long long rank = ((unsigned long long)ci << 30) | ((unsigned long long)cj << 20) |
                                    ((unsigned long long)(y - ci * hcell) << 10) | (unsigned long long)(x - cj * wcell);
    return ((unsigned long long)(e >> 24) << 40) | (0xFFFFFFFFFFull - rank);
}

// MAXN = node-list capacity (>= the largest per-level quota + 3), CL = keys kept in LDS; a block handles its (frame, level) iff the candidate
// count nk lies in its class: nk_lo < nk <= CL (KEYS_IN_LDS) or nk > nk_lo (global keys).  Small classes leave room for several workgroups per CU.
template <bool KEYS_IN_LDS, int MAXN, int CL>
SGX_KERNEL(512) k_octree(SgxOrbGeom g, const uint32_t *cand, const int *cand_count, uint16_t *node_scratch,
                                     uint32_t *sel, int *sel_count, uint32_t *status, int nk_lo)
{
    // keys: packed x | y<<12 | S<<24 and the list position of the node that owns each key.
    // Normal case: both in LDS.  A (frame, level) with more than SGX_CAND_LDS candidates (e.g. pure
    // noise) is handled by the KEYS_IN_LDS=false instantiation, which reads the keys from the
    // candidate buffer and keeps node_of in a global scratch slice; each block runs in exactly one.
    SGX_LDS uint32_t kxy_lds[KEYS_IN_LDS ? CL : 1];
    SGX_LDS uint16_t node_lds[KEYS_IN_LDS ? CL : 1];
    // node list (array in list order), ping-pong
    SGX_LDS SgxOctNode nb[2][MAXN];
    SGX_LDS uint16_t ncnt[2][MAXN];      // keys per node
    SGX_LDS uint16_t nseq[2][MAXN];      // creation index within the pass that created it
    SGX_LDS uint8_t nflag[2][MAXN];      // 1: created by the last pass with >1 key (the reference's vSizeAndPointerToNode)
    // per-pass scratch indexed by OLD list position
    SGX_LDS int quad[MAXN][4];           // keys per quadrant; later the children's new positions ([0] for kept nodes)
    SGX_LDS int scanA[MAXN];             // children created before (in creation order)
    SGX_LDS int scanB[MAXN];             // kept nodes before (in list order)
    SGX_LDS uint8_t splitting[MAXN];
    SGX_LDS uint16_t order[MAXN];        // phase 2: processing rank -> old position
    SGX_LDS uint16_t rank_of[MAXN];
    SGX_LDS unsigned long long best[MAXN];
    SGX_LDS int s_size, s_C, s_kept, s_ntoexpand, s_stop, s_m, s_overflow;

    // grid = (levels, frames) by default; (frames, levels) — level 0, the longest-running, dispatched first — is a tuning tap (sgx_orb.cpp)
    const bool frames_fast = (int)gridDim.y == g.nlevels;            // the two grid shapes are told apart by their extent
    const int frame = (int)(frames_fast ? blockIdx.x : blockIdx.y), level = (int)(frames_fast ? blockIdx.y : blockIdx.x);
    const SgxLevel L = g.lv[level];
    const int N = L.quota;
    int nk = cand_count[frame * g.nlevels + level];
    if (nk > L.cand_cap) nk = L.cand_cap;
    if (nk <= nk_lo || (KEYS_IN_LDS && nk > CL)) return;
    const uint32_t *src = cand + (size_t)frame * g.cand_pitch + L.cand_off;
    uint32_t *kxy = KEYS_IN_LDS ? kxy_lds : (uint32_t *)src;
    uint16_t *node_of = KEYS_IN_LDS ? node_lds : node_scratch + (size_t)frame * g.cand_pitch + L.cand_off;
    uint32_t *out = sel + ((size_t)frame * g.nlevels + level) * SGX_OCT_MAXN;
    const int NT = (int)blockDim.x;

    const int minX = SGX_BORDER, maxX = L.w - SGX_EDGE + 3, minY = SGX_BORDER, maxY = L.h - SGX_EDGE + 3;
    const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));       // :544
    const float hX = nIni > 0 ? (float)(maxX - minX) / (float)nIni : 1.f;             // :546

    if (nk == 0 || nIni < 1 || nIni > 64) {
        SGX_THREADS_BEGIN(tid) if (tid == 0) sel_count[frame * g.nlevels + level] = 0; SGX_THREADS_END
        return;
    }

    // ---- roots (:553-571); empty roots are dropped (:573-586)
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < nIni; i += NT) scanA[i] = 0;
    if (tid == 0) s_overflow = 0;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < nk; k += NT) {
        const uint32_t e = src[k];
        if (KEYS_IN_LDS) kxy[k] = e;
        const int r = (int)((float)(e & 0xFFF) / hX);       // vpIniNodes[kp.pt.x/hX] :570
        node_of[k] = (uint16_t)r;
        sgx_atomic_add(&scanA[r], 1);
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) {     // nIni is tiny: compact the non-empty roots serially
        int n = 0;
        for (int i = 0; i < nIni; i++) {
            const int c = scanA[i];
            if (c == 0) { scanB[i] = -1; continue; }
            nb[0][n].ulx = (uint16_t)(int)(hX * (float)i); nb[0][n].uly = 0;
            nb[0][n].urx = (uint16_t)(int)(hX * (float)(i + 1)); nb[0][n].bry = (uint16_t)(maxY - minY);
            ncnt[0][n] = (uint16_t)c; nseq[0][n] = (uint16_t)i; nflag[0][n] = 0;
            scanB[i] = n; n++;
        }
        s_size = n;
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < nk; k += NT) node_of[k] = (uint16_t)scanB[node_of[k]];
    SGX_THREADS_END
    SGX_SYNC();

    int cur = 0;            // ping-pong index of the live node arrays
    int phase2 = 0;
    for (int guard = 0; guard < 4096; guard++) {
        const int size = s_size;
        const int nxt = cur ^ 1;
        // ---- choose the (provisional) splitting set
        if (!phase2) {
            SGX_THREADS_BEGIN(tid)
            for (int i = tid; i < size; i += NT) splitting[i] = ncnt[cur][i] > 1;       // !bNoMore
            if (tid == 0) { s_stop = 0x7FFFFFFF; s_m = 0; }
            SGX_THREADS_END
        } else {
            // processing rank by (count, creation order) descending (:685-686)
            SGX_THREADS_BEGIN(tid)
            for (int i = tid; i < size; i += NT) {
                int r = -1;
                if (nflag[cur][i]) {
                    const uint32_t ki = ((uint32_t)ncnt[cur][i] << 16) | nseq[cur][i];
                    r = 0;
                    for (int j = 0; j < size; j++)
                        if (nflag[cur][j]) { const uint32_t kj = ((uint32_t)ncnt[cur][j] << 16) | nseq[cur][j]; r += kj > ki; }
                    order[r] = (uint16_t)i;
                }
                rank_of[i] = (uint16_t)r;
                splitting[i] = r >= 0;
            }
            SGX_THREADS_END
        }
        SGX_SYNC();
        // ---- quadrant populations of the splitting nodes
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < size; i += NT) { quad[i][0] = 0; quad[i][1] = 0; quad[i][2] = 0; quad[i][3] = 0; }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int k = tid; k < nk; k += NT) {
            const int p = node_of[k];
            if (splitting[p]) sgx_atomic_add(&quad[p][sgx_oct_quadrant(nb[cur][p], kxy[k])], 1);
        }
        SGX_THREADS_END
        SGX_SYNC();
        // ---- phase 2: stop at the first split that makes the list reach N (:731-732)
        if (phase2) {
            SGX_THREADS_BEGIN(tid)
            if (tid == 0) {
                int m = 0;
                for (int i = 0; i < size; i++) m += nflag[cur][i];
                int sz = size, stop = 0x7FFFFFFF;
                for (int r = 0; r < m; r++) {
                    const int i = order[r];
                    sz += (quad[i][0] > 0) + (quad[i][1] > 0) + (quad[i][2] > 0) + (quad[i][3] > 0) - 1;
                    if (sz >= N) { stop = r; break; }
                }
                s_stop = stop; s_m = m;
            }
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            for (int i = tid; i < size; i += NT) if (splitting[i] && (int)rank_of[i] > s_stop) splitting[i] = 0;
            SGX_THREADS_END
            SGX_SYNC();
        }
        // ---- children per creation slot (phase 1: list order, phase 2: processing rank) and kept flags
        SGX_THREADS_BEGIN(tid)
        for (int s = tid; s < size; s += NT) {
            int i = s;
            bool on = splitting[s];
            if (phase2) { on = s < s_m && s <= s_stop; i = on ? (int)order[s] : 0; }
            scanA[s] = on ? (quad[i][0] > 0) + (quad[i][1] > 0) + (quad[i][2] > 0) + (quad[i][3] > 0) : 0;
            scanB[s] = splitting[s] ? 0 : 1;
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        sgx_block_exclusive_scan_i32(scanA, size, &s_C, tid);
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        sgx_block_exclusive_scan_i32(scanB, size, &s_kept, tid);
        if (tid == 0) s_ntoexpand = 0;
        SGX_THREADS_END
        SGX_SYNC();
        const int C = s_C;
        const int new_size = C + s_kept;
        if (new_size > MAXN) {
            SGX_THREADS_BEGIN(tid) if (tid == 0) { sgx_atomic_or(status, SGX_ST_NODE_OVERFLOW); s_overflow = 1; } SGX_THREADS_END
            SGX_SYNC();
            break;
        }
        // ---- build the new list: reverse(children in creation order) ++ kept nodes in old order
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < size; i += NT) {
            if (splitting[i]) {
                const SgxOctNode b = nb[cur][i];
                const int halfX = (int)ceilf((float)(b.urx - b.ulx) / 2), halfY = (int)ceilf((float)(b.bry - b.uly) / 2);
                int ci = scanA[phase2 ? (int)rank_of[i] : i];
                for (int q = 0; q < 4; q++) {
                    const int c = quad[i][q];
                    if (c == 0) continue;
                    const int pos = C - 1 - ci;
                    SgxOctNode nn;
                    nn.ulx = (q & 1) ? (uint16_t)(b.ulx + halfX) : b.ulx;
                    nn.urx = (q & 1) ? b.urx : (uint16_t)(b.ulx + halfX);
                    nn.uly = (q & 2) ? (uint16_t)(b.uly + halfY) : b.uly;
                    nn.bry = (q & 2) ? b.bry : (uint16_t)(b.uly + halfY);
                    nb[nxt][pos] = nn; ncnt[nxt][pos] = (uint16_t)c; nseq[nxt][pos] = (uint16_t)ci;
                    nflag[nxt][pos] = c > 1;
                    if (c > 1) sgx_atomic_add(&s_ntoexpand, 1);
                    quad[i][q] = pos;
                    ci++;
                }
            } else {
                const int pos = C + scanB[i];
                nb[nxt][pos] = nb[cur][i]; ncnt[nxt][pos] = ncnt[cur][i]; nseq[nxt][pos] = nseq[cur][i];
                nflag[nxt][pos] = 0;
                quad[i][0] = pos;
            }
        }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        for (int k = tid; k < nk; k += NT) {
            const int p = node_of[k];
            node_of[k] = (uint16_t)(splitting[p] ? quad[p][sgx_oct_quadrant(nb[cur][p], kxy[k])] : quad[p][0]);
        }
        if (tid == 0) s_size = new_size;
        SGX_THREADS_END
        SGX_SYNC();
        cur = nxt;
        // ---- termination (:670-675, :735-736)
        if (new_size >= N || new_size == size) break;
        if (!phase2 && new_size + s_ntoexpand * 3 > N) phase2 = 1;
    }

    // ---- retain the best key of every node (:745-761) and emit in list order
    const int fsize = s_size;
    SGX_THREADS_BEGIN(tid)
    for (int i = tid; i < fsize; i += NT) best[i] = 0ull;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < nk; k += NT) sgx_atomic_max(&best[node_of[k]], sgx_oct_pick_key(kxy[k], L.wcell, L.hcell));
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int k = tid; k < nk; k += NT)
        if (best[node_of[k]] == sgx_oct_pick_key(kxy[k], L.wcell, L.hcell)) out[node_of[k]] = kxy[k];   // the key is unique per pixel
    if (tid == 0) sel_count[frame * g.nlevels + level] = s_overflow ? 0 : fsize;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// sinf/cosf exactly as the host libm computes them for the reference (`cos(float)` / `sin(float)`
// at ORBextractor.cc:114 resolve to glibc cosf/sinf).  Restatement of glibc >= 2.28
// sysdeps/ieee754/flt-32/{s_sinf.c,s_cosf.c,sincosf.h} (double-precision polynomial, 2^24-scaled
// quadrant reduction); checked bit-for-bit against glibc 2.35 on every float in [0, 2*pi*1.001]
// (tests/test_sincosf.py samples it).  Valid for |x| < 120, which covers angle*pi/180, angle in [0,360).
// ---------------------------------------------------------------------------------------------
SGX_DEV float sgx_sincos_poly(double x, double x2, int neg_cos, int n)
{
    const double c0 = neg_cos ? -0x1p0 : 0x1p0, c1 = neg_cos ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = neg_cos ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5, c3 = neg_cos ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = neg_cos ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2, sp = s2 + x2 * s3, x7 = x3 * x2, s = x + x3 * s1;
        return (float)(s + x7 * sp);
    }
    const double x4 = x2 * x2, cp2 = c3 + x2 * c4, cp1 = c0 + x2 * c1, x6 = x4 * x2, c = cp1 + x4 * c2;
    return (float)(c + x6 * cp2);
}

SGX_DEV void sgx_sincosf(float y, float *sn, float *cs)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    double x = (double)y;
    uint32_t u; memcpy(&u, &y, 4);
    const uint32_t top = (u >> 20) & 0x7ff;
    if (top < 0x3f4u) {                        // |y| < pi/4   (abstop12(0x1.921FB6p-1f) = 0x3f4)
        const double x2 = x * x;
        if (top < 0x398u) { *sn = y; *cs = 1.0f; return; }      // |y| < 2^-12
        *sn = sgx_sincos_poly(x, x2, 0, 0);
        *cs = sgx_sincos_poly(x, x2, 0, 1);
        return;
    }
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * hpi;
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const int neg = (n & 2) != 0;
    *sn = sgx_sincos_poly(x * sgn, x * x, neg, n);
    *cs = sgx_sincos_poly(x * sgn, x * x, neg, n ^ 1);
}

// cv::fastAtan2 (degrees) — OpenCV 3.4 atan_f32 polynomial; see oracle/orb_oracle.c orc_fast_atan2
SGX_DEV float sgx_fast_atan2(float y, float x)
{
    const float k = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + eps); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + eps); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// (hi:lo) >> (8*sh) as 32 bits, sh in 0..3 (v_alignbyte_b32)

SGX_DEV int sgx_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}

// ---------------------------------------------------------------------------------------------
// k_orient_desc: one 64-lane wave per output keypoint.
//   IC_Angle (ORBextractor.cc:78-105) on the un-blurred level,
//   GaussianBlur 7x7 sigma 2 REFLECT_101 (:1086-1087; OpenCV 3.4.15 fixed-point: taps
//   {18,34,48,56,48,34,18}/256, 8.8 horizontal, 16.16 vertical, +0.5 round) evaluated only on the
//   37x37 window the steered pattern can reach (|offset| <= round(18.38)),
//   computeOrbDescriptor (:109-148), keypoint scaling/packing (:838-848, :1096-1104).
// Output keypoints of a frame are ordered level 0..nlevels-1, inside a level in octree list order.
// ---------------------------------------------------------------------------------------------
#define SGX_PR 21                  /* source patch radius: 18 (pattern reach) + 3 (blur) */
#define SGX_PW (2 * SGX_PR + 1)    /* 43 */
#define SGX_BR 18
#define SGX_BW (2 * SGX_BR + 1)    /* 37 */
#define SGX_PS 48                  /* LDS row stride of the staged patch (bytes) */
#define SGX_HS 40                  /* LDS row stride of the horizontal-pass buffer (u16) */

#ifdef SGX_DEBUG_TAPS      /* the first design (blur of a 37 x 37 window per keypoint); superseded by k_blur_levels + k_orient_desc4; tap build only: SGX_TUNE_ORB_PATCH_BLUR */
SGX_KERNEL(64) k_orient_desc(SgxOrbGeom g, const uint8_t *gray, int gray_pitch, const uint8_t *pyr,
                             const uint32_t *sel, const int *sel_count, unsigned long long umax_packed, const signed char *pattern,
                             uint8_t *kps_raw, uint8_t *desc, int *count, int cap, int batch, uint32_t *status)
{
    SGX_LDS uint32_t patch_dw[SGX_PW * SGX_PS / 4];      // 43 rows x 48 bytes; column c of the patch sits at byte lead + c
    uint8_t *patch = (uint8_t *)patch_dw;
    SGX_LDS uint32_t hbuf_dw[SGX_PW * SGX_HS / 2];       // horizontal pass, 8.8 fixed point, row stride SGX_HS u16
    uint16_t *hbuf = (uint16_t *)hbuf_dw;
    SGX_LDS uint8_t blur[SGX_BW * (SGX_BW + 1)];
    SGX_LDS uint8_t bits[256];
    SGX_LDS int s_m01, s_m10;
    SGX_LDS float s_a, s_b;
    const int GK0 = 18, GK1 = 34, GK2 = 48, GK3 = 56;

    // block -> (slot, frame).  1-D grid of kp_cap * batch blocks; the hardware places block n on XCD n % 8, so frames are dealt
    // to XCDs (frame f -> XCD f % 8) and each XCD walks its frames one after the other: a frame's pyramid (1.2 MB) stays in
    // that XCD's 4 MB L2 while its ~1000 patches are read (speed only; any placement is correct).
    int slot, frame;
    {
        const int n = (int)blockIdx.x, kc = g.kp_cap;
        if ((batch & 7) == 0) { const int xcd = n & 7, j = n >> 3; frame = xcd + 8 * (j / kc); slot = j % kc; }
        else { frame = n / kc; slot = n % kc; }
    }
    int level = -1, base = 0, total = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int n = sel_count[frame * g.nlevels + l];
        if (level < 0 && slot < total + n) { level = l; base = total; }
        total += n;
    }
    if (slot == 0) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) { count[frame] = total < cap ? total : cap; if (total > cap) sgx_atomic_or(status, SGX_ST_KP_OVERFLOW); }
        SGX_THREADS_END
    }
    if (level < 0 || slot >= cap) return;

    const SgxLevel L = g.lv[level];
    const uint32_t e = sel[((size_t)frame * g.nlevels + level) * SGX_OCT_MAXN + (slot - base)];
    const int kx = (int)(e & 0xFFF) + SGX_BORDER, ky = (int)((e >> 12) & 0xFFF) + SGX_BORDER;   // :844-845
    int stride;
    const uint8_t *img = sgx_level_ptr(g, gray, gray_pitch, pyr, frame, level, &stride);

    // stage the 43x43 source patch.  Fast path (patch inside the image): aligned dword loads, one LDS dword store each;
    // near the border: byte loads with BORDER_REFLECT_101 indices.
    const int px0 = kx - SGX_PR, py0 = ky - SGX_PR;
    const bool inside = px0 >= 0 && py0 >= 0 && ky + SGX_PR < L.h && kx + SGX_PR < L.w && kx + SGX_PR + 4 < stride;
    const int lead = inside ? (px0 & 3) : 0;
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { s_m01 = 0; s_m10 = 0; }
    if (inside) {
        const int xa = px0 - lead;
#pragma unroll
        for (int it_ = 0; it_ < (SGX_PW * (SGX_PS / 4) + 63) / 64; it_++) { const int i = tid + 64 * it_; if (i >= SGX_PW * (SGX_PS / 4)) break;
            const int r = i / (SGX_PS / 4), q = i - r * (SGX_PS / 4);
            if (4 * q < lead + SGX_PW) patch_dw[i] = *(const uint32_t *)(img + (size_t)(py0 + r) * stride + xa + 4 * q);
        }
    } else {
        for (int i = tid; i < SGX_PW * SGX_PW; i += 64) {
            const int r = i / SGX_PW, c = i - r * SGX_PW;
            const int yy = sgx_reflect101(py0 + r, L.h), xx = sgx_reflect101(px0 + c, L.w);
            patch[r * SGX_PS + c] = img[(size_t)yy * stride + xx];
        }
    }
    SGX_THREADS_END
    SGX_SYNC();

    // intensity-centroid moments over the radius-15 disc (umax rows), integer exact in any order
    SGX_THREADS_BEGIN(tid)
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int it_ = 0; it_ < (31 * 31 + 63) / 64; it_++) { const int i = tid + 64 * it_; if (i >= 31 * 31) break;
        const int v = i / 31 - 15, u = i - (v + 15) * 31 - 15;
        const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
        if (au <= (int)((umax_packed >> (4 * av)) & 15ull)) {             // umax[av], 16 x 4 bits packed by the host
            const int I = patch[(SGX_PR + v) * SGX_PS + lead + SGX_PR + u];
            m10 += u * I; m01 += v * I;
        }
    }
    sgx_atomic_add(&s_m10, m10); sgx_atomic_add(&s_m01, m01);
    // horizontal blur pass (8.8 fixed point): task = (row, 8-column segment); the 14 source bytes come from 5 aligned dword
    // reads realigned with v_alignbyte, every byte is read once (sliding window in registers)
#pragma unroll
    for (int it_ = 0; it_ < (SGX_PW * 5 + 63) / 64; it_++) { const int t = tid + 64 * it_; if (t >= SGX_PW * 5) break;
        const int r = t / 5, sg = t - r * 5;
        const uint32_t *w = patch_dw + (r * SGX_PS + 8 * sg) / 4;
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = (8 * sg + 16 < SGX_PS) ? w[4] : 0u;
        const uint32_t v0 = sgx_alignbyte(w1, w0, lead), v1 = sgx_alignbyte(w2, w1, lead), v2 = sgx_alignbyte(w3, w2, lead), v3 = sgx_alignbyte(w4, w3, lead);
        int b[16];
#pragma unroll
        for (int j = 0; j < 4; j++) { b[j] = (v0 >> (8 * j)) & 255; b[4 + j] = (v1 >> (8 * j)) & 255; b[8 + j] = (v2 >> (8 * j)) & 255; b[12 + j] = (v3 >> (8 * j)) & 255; }
        uint32_t o[4];
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            const uint32_t h0 = (uint32_t)(GK0 * (b[c] + b[c + 6]) + GK1 * (b[c + 1] + b[c + 5]) + GK2 * (b[c + 2] + b[c + 4]) + GK3 * b[c + 3]);
            const uint32_t h1 = (uint32_t)(GK0 * (b[c + 1] + b[c + 7]) + GK1 * (b[c + 2] + b[c + 6]) + GK2 * (b[c + 3] + b[c + 5]) + GK3 * b[c + 4]);
            o[c >> 1] = h0 | (h1 << 16);
        }
        uint32_t *dst = hbuf_dw + (r * SGX_HS + 8 * sg) / 2;          // columns >= 37 of the last segment are never read
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
    }
    SGX_THREADS_END
    SGX_SYNC();

    SGX_THREADS_BEGIN(tid)
    if (tid == 63) {         // the last lane has one blur task fewer than the others (185 = 2*64 + 57)
        const float angle = sgx_fast_atan2((float)s_m01, (float)s_m10);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float sn, cs;
        sgx_sincosf(angle * factorPI, &sn, &cs);
        s_a = cs; s_b = sn;
        // keypoint record (cv::KeyPoint layout)
        float *kp = (float *)(kps_raw + ((size_t)frame * cap + slot) * 28);
        float fx = (float)kx, fy = (float)ky;
        if (level != 0) { fx = fx * L.scale; fy = fy * L.scale; }
        kp[0] = fx; kp[1] = fy; kp[2] = (float)L.patch_size; kp[3] = angle; kp[4] = (float)(e >> 24);
        ((int *)kp)[5] = level; ((int *)kp)[6] = -1;
    }
    // vertical blur pass (16.16) + rounding: task = (column, 8-row segment), 14 sliding reads per 8 outputs
#pragma unroll
    for (int it_ = 0; it_ < (SGX_BW * 5 + 63) / 64; it_++) { const int t = tid + 64 * it_; if (t >= SGX_BW * 5) break;
        const int sg = t / SGX_BW, c = t - sg * SGX_BW;
        const int r0 = 8 * sg, nr = (SGX_BW - r0) < 8 ? (SGX_BW - r0) : 8;
        uint32_t h[14];
#pragma unroll
        for (int j = 0; j < 14; j++) h[j] = (r0 + j < SGX_PW) ? (uint32_t)hbuf[(r0 + j) * SGX_HS + c] : 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j < nr) {
                const uint32_t acc = (uint32_t)GK0 * (h[j] + h[j + 6]) + (uint32_t)GK1 * (h[j + 1] + h[j + 5]) + (uint32_t)GK2 * (h[j + 2] + h[j + 4]) + (uint32_t)GK3 * h[j + 3];
                blur[(r0 + j) * (SGX_BW + 1) + c] = (uint8_t)((acc + 32768u) >> 16);
            }
        }
    }
    SGX_THREADS_END
    SGX_SYNC();

    SGX_THREADS_BEGIN(tid)
    const float a = s_a, b = s_b;
#pragma unroll
    for (int it_ = 0; it_ < 4; it_++) { const int t = tid + 64 * it_;
        const uint32_t pw = *(const uint32_t *)(pattern + 4 * t);          // one test = 4 signed bytes (x0, y0, x1, y1)
        const float x0 = (float)(signed char)(pw & 255u), y0 = (float)(signed char)((pw >> 8) & 255u), x1 = (float)(signed char)((pw >> 16) & 255u), y1 = (float)(signed char)(pw >> 24);
        const int r0 = sgx_cvround(x0 * b + y0 * a), c0 = sgx_cvround(x0 * a - y0 * b);
        const int r1 = sgx_cvround(x1 * b + y1 * a), c1 = sgx_cvround(x1 * a - y1 * b);
        const int t0 = blur[(SGX_BR + r0) * (SGX_BW + 1) + SGX_BR + c0];
        const int t1 = blur[(SGX_BR + r1) * (SGX_BW + 1) + SGX_BR + c1];
        bits[t] = (uint8_t)(t0 < t1);
    }
    SGX_THREADS_END
    SGX_SYNC();

    SGX_THREADS_BEGIN(tid)
    if (tid < 32) {
        int v = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) v |= bits[8 * tid + j] << j;
        desc[((size_t)frame * cap + slot) * 32 + tid] = (uint8_t)v;
    }
    SGX_THREADS_END
}
#endif      /* SGX_DEBUG_TAPS */

// ---------------------------------------------------------------------------------------------
// k_blur_levels: GaussianBlur(level, 7x7, sigma 2, BORDER_REFLECT_101) of every pyramid level (ORBextractor.cc:1086-1087), whole levels like the
// reference, as one launch over 64 x 32 tiles: the tile + 3-px halo is staged in LDS (aligned dwords inside the image, reflected byte loads on
// border tiles), horizontal pass in 8.8 fixed point and vertical pass in 16.16 with sliding windows (every staged value is read once per pass), exact
// OpenCV taps {18,34,48,56,48,34,18}/256 and rounding.  It costs ~20 VALU operations per pixel, less than half of blurring a 37x37 window per
// keypoint, and leaves k_orient_desc2 with plain gathers.  block -> (tile, frame), frame fastest (XCD-aware like k_fast_cells).
// ---------------------------------------------------------------------------------------------
struct SgxBlurTile { short level, x0, y0, w, h, pad0, pad1, pad2; };
#define SGX_BT_W 64
#define SGX_BT_H 58            /* + 6 halo rows = 64 staged rows: the horizontal pass is exactly two rounds of 256 (row, 8-column) tasks */
#define SGX_BT_IS 80            /* LDS row stride of the staged input (bytes): 3 lead + 3 + 64 + 3, dword aligned */
#define SGX_BT_HS 66            /* COLUMN stride of the horizontal-pass buffer (u16): it is stored transposed, [column][row], 64 staged rows + 2 (33 dwords: odd -> the columns of a wave fall on different banks) */

// Persistent and software-pipelined: a workgroup walks a list of tiles; the global loads of the NEXT tile are issued into registers right after the
// horizontal pass (the staged input is dead from then on) and land in LDS after the vertical pass, so a tile costs its two passes and three barriers, not a load round trip
// on top (the one-tile-per-workgroup form ran at 32 waves per CU that mostly waited: 6.8 us per tile).
// Which tiles (round 4).  grid = parts x batch; workgroup b owns frame b % batch (XCD b % 8 = frame % 8 for batches that are multiples of 8: a frame's pyramid stays in one L2)
// and the CONTIGUOUS tile range [k T / parts, (k + 1) T / parts) of that frame, k = b / batch, in table order = row-major inside a level.  A 64-byte-wide tile + halo straddles
// two 128-byte lines, so x-neighbours share every second line: walking them back to back finds the shared line still in L2.  Round 3 walked tile = k, k + parts, k + 2 parts, ...
// per workgroup: the x-neighbours of a tile were then other workgroups' work at unrelated times, every tile fetched its two lines per row from HBM and the kernel read 1 876 MB
// per 512 frames for 487 MB of pyramid (one tile per workgroup in (tile, frame) order reads 520 MB but gives up the pipelining).
#define SGX_BT_PRE 5            /* staged dwords per thread: 64 rows x 20 dwords / 256 threads (the smallest workgroup the host launches) */
SGX_KERNEL(512) k_blur_levels(SgxOrbGeom g, const SgxBlurTile *tiles, const uint8_t *gray, int gray_pitch, const uint8_t *pyr, uint8_t *blur, int batch)
{
    SGX_LDS uint32_t in_dw[(SGX_BT_H + 6) * SGX_BT_IS / 4];
    SGX_LDS uint32_t h_dw[SGX_BT_W * SGX_BT_HS / 2 + 8];           // + the 8-dword read of the last column's last segment
    SGX_LDS uint32_t o_dw[SGX_BT_H * SGX_BT_W / 4];
    SGX_PRIV_DECL(uint32_t, pre, SGX_BT_PRE, 512);
    uint8_t *in = (uint8_t *)in_dw; uint16_t *hb = (uint16_t *)h_dw; uint8_t *ob = (uint8_t *)o_dw;
    const int total = g.nblur_tiles * batch;
    // staging: always aligned dwords.  Rows outside the level are fetched from their BORDER_REFLECT_101 source row; dwords that would start outside the row
    // are clamped into it (their bytes are garbage) and the at most 3 + 3 halo columns that lie outside the level are then copied from their reflected
    // columns, which are always staged.  (A per-byte reflected loader for border tiles — 30 % of the tiles — cost more than the two blur passes.)
#define SGX_BLUR_FETCH(F_, T_, DST_)                                                                                                       \
    {                                                                                                                                     \
        const int f_ = (F_); const SgxBlurTile t_ = tiles[(T_)]; const SgxLevel L_ = g.lv[t_.level]; int st_;                             \
        const uint8_t *img_ = sgx_level_ptr(g, gray, gray_pitch, pyr, f_, t_.level, &st_);                                                \
        const int rows_ = t_.h + 6, xs_ = t_.x0 - 3, ys_ = t_.y0 - 3, xa_ = xs_ - (xs_ & 3), maxq_ = (st_ >> 2) - 1;                      \
        for (int u_ = 0; u_ < SGX_BT_PRE; u_++) {                                                                                         \
            const int i_ = tid + u_ * (int)blockDim.x;                                                                                    \
            if (i_ < rows_ * (SGX_BT_IS / 4)) {                                                                                           \
                const int r_ = i_ / (SGX_BT_IS / 4), q_ = i_ - r_ * (SGX_BT_IS / 4);                                                      \
                int yy_ = ys_ + r_; yy_ = yy_ < 0 ? -yy_ : (yy_ >= L_.h ? 2 * (L_.h - 1) - yy_ : yy_);                                    \
                const int dq_ = min(max((xa_ >> 2) + q_, 0), maxq_);                                                                      \
                DST_[u_] = ((const uint32_t *)(img_ + (size_t)yy_ * st_))[dq_];                                                           \
            }                                                                                                                             \
        }                                                                                                                                 \
    }
    if ((int)blockIdx.x >= total) return;
    const int parts = (int)gridDim.x / batch, part = (int)blockIdx.x / batch;              // host: gridDim.x = parts * batch, 1 <= parts <= tiles per frame
    const int cur_f = (int)blockIdx.x - part * batch;
    const int t_end = (int)(((long long)(part + 1) * g.nblur_tiles) / parts);
    int cur_t = (int)(((long long)part * g.nblur_tiles) / parts);
    SGX_THREADS_BEGIN(tid)
    SGX_PRIV_BIND(pre, tid);
    SGX_BLUR_FETCH(cur_f, cur_t, pre)
    for (int u = 0; u < SGX_BT_PRE; u++) { const int i = tid + u * (int)blockDim.x; if (i < (SGX_BT_H + 6) * (SGX_BT_IS / 4)) in_dw[i] = pre[u]; }
    SGX_THREADS_END
    SGX_SYNC();
    for (; cur_t < t_end; cur_t++) {
        const int frame = cur_f, nxt_f = cur_f, nxt_t = cur_t + 1;
        const bool more = nxt_t < t_end;
        const SgxBlurTile t = tiles[cur_t];
        const SgxLevel L = g.lv[t.level];
        const int rows = t.h + 6, xs = t.x0 - 3;
        const int lead = xs & 3;
        if (xs < 0 || t.x0 + t.w + 3 > L.w) {
            SGX_THREADS_BEGIN(tid)
            for (int k = tid; k < rows * 6; k += (int)blockDim.x) {
                const int r = k / 6, hc = k - r * 6;
                const int x = hc < 3 ? xs + hc : t.x0 + t.w + (hc - 3);               // level column of this halo slot
                if (x < 0 || x >= L.w) {
                    const int xr = x < 0 ? -x : 2 * (L.w - 1) - x;
                    in[r * SGX_BT_IS + lead + (x - xs)] = in[r * SGX_BT_IS + lead + (xr - xs)];
                }
            }
            SGX_THREADS_END
            SGX_SYNC();
        }
        // horizontal pass: task = (row, 8-column segment).  The 7 taps (18 34 48 56 48 34 18) / 256 are exact u8 weights: an output is TWO v_dot4_u32_u8 on byte windows
        // (columns c .. c+3 against 18 34 48 56, columns c+4 .. c+7 against 48 34 18 0) cut out of the 16 staged bytes with v_alignbyte — 9 realignments + 16 dot products
        // for 8 outputs instead of 16 byte extractions + 56 multiply-adds.  The 8.8 results are stored TRANSPOSED ([column][row], u16) so that the vertical pass finds the
        // rows of a column as packed pairs.
        SGX_THREADS_BEGIN(tid)
        for (int k = tid; k < rows * (SGX_BT_W / 8); k += (int)blockDim.x) {
            const int r = k >> 3, sg = k & 7;
            if (8 * sg >= t.w) continue;
            const uint32_t *w = in_dw + (r * SGX_BT_IS + 8 * sg) / 4;
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
            uint32_t A[12];
            A[0] = sgx_alignbyte(w1, w0, lead); A[4] = sgx_alignbyte(w2, w1, lead); A[8] = sgx_alignbyte(w3, w2, lead);
            const uint32_t v3 = sgx_alignbyte(w4, w3, lead);
    #pragma unroll
            for (int q = 1; q < 4; q++) { A[q] = sgx_alignbyte(A[4], A[0], q); A[4 + q] = sgx_alignbyte(A[8], A[4], q); A[8 + q] = sgx_alignbyte(v3, A[8], q); }
            const uint32_t WLO = 18u | (34u << 8) | (48u << 16) | (56u << 24), WHI = 48u | (34u << 8) | (18u << 16);
    #pragma unroll
            for (int c = 0; c < 8; c++) hb[(8 * sg + c) * SGX_BT_HS + r] = (uint16_t)sgx_udot4(A[c], WLO, sgx_udot4(A[c + 4], WHI, 0u));      // <= 255 * 256: fits 16 bits
        }
        SGX_THREADS_END
        // the next tile's loads leave now; their latency hides behind the vertical pass
        SGX_THREADS_BEGIN(tid)
        SGX_PRIV_BIND(pre, tid);
        if (more) SGX_BLUR_FETCH(nxt_f, nxt_t, pre)
        SGX_THREADS_END
        SGX_SYNC();
        // vertical pass: task = (column, 8-row segment).  Rows r0 .. r0+15 of the column are 8 aligned dwords = the row pairs starting at even rows; the odd-start pairs come
        // from v_alignbyte; an output is FOUR v_dot2_u32_u16 (pairs j, j+2, j+4 against (18,34) (48,56) (48,34), pair j+6 against (18,0)) chained through the accumulator that
        // starts at the rounding constant.  Exact integers: the bytes are those of the multiply-add version.
        SGX_THREADS_BEGIN(tid)
        for (int k = tid; k < 64 * ((SGX_BT_H + 7) / 8); k += (int)blockDim.x) {
            const int c = k & 63, sg = k >> 6, r0 = 8 * sg;
            if (c < t.w && r0 < t.h) {
                const uint32_t *col = h_dw + (c * SGX_BT_HS + r0) / 2;
                uint32_t P[14];                                               // P[j] = (h[r0 + j], h[r0 + j + 1])
    #pragma unroll
                for (int q = 0; q < 7; q++) P[2 * q] = col[q];
                const uint32_t last = col[7];
    #pragma unroll
                for (int q = 0; q < 6; q++) P[2 * q + 1] = sgx_alignbyte(P[2 * q + 2], P[2 * q], 2);
                P[13] = sgx_alignbyte(last, P[12], 2);
                const uint32_t W0 = 18u | (34u << 16), W1 = 48u | (56u << 16), W2 = 48u | (34u << 16), W3 = 18u;
    #pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t acc = sgx_udot2(P[j], W0, sgx_udot2(P[j + 2], W1, sgx_udot2(P[j + 4], W2, sgx_udot2(P[j + 6], W3, 32768u))));
                    if (r0 + j < SGX_BT_H) ob[(r0 + j) * SGX_BT_W + c] = (uint8_t)(acc >> 16);
                }
            }
        }
        SGX_THREADS_END
        SGX_THREADS_BEGIN(tid)
        SGX_PRIV_BIND(pre, tid);
        if (more) for (int u = 0; u < SGX_BT_PRE; u++) { const int i = tid + u * (int)blockDim.x; if (i < (SGX_BT_H + 6) * (SGX_BT_IS / 4)) in_dw[i] = pre[u]; }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        uint8_t *dst = blur + (size_t)frame * g.blur_pitch + L.boff;
        for (int i = tid; i < t.h * (SGX_BT_W / 4); i += (int)blockDim.x) {
            const int r = i >> 4, q = i & 15;
            if (4 * q < t.w) *(uint32_t *)(dst + (size_t)(t.y0 + r) * L.bstride + t.x0 + 4 * q) = o_dw[i];        // rows are padded to 64 bytes: the last dword may spill into the padding
        }
        SGX_THREADS_END
    }
#undef SGX_BLUR_FETCH
}

// ---------------------------------------------------------------------------------------------
// k_orient_desc2: IC_Angle (ORBextractor.cc:78-105) on the raw level + steered BRIEF (:109-148) sampled from the blurred level written by k_blur_levels.
// One wave per keypoint slot, XCD-aware block order as k_orient_desc.  Keypoints sit >= 19 px from the border (EDGE_THRESHOLD) and the rotated pattern
// reaches 18 px, so neither the 31x31 moment patch nor the samples ever leave the level: no border path.
// ---------------------------------------------------------------------------------------------
#define SGX_MS 36              /* LDS row stride of the staged 31x31 moment patch (bytes) */
#ifdef SGX_DEBUG_TAPS      /* superseded by k_orient_desc4 (four keypoints per wave); tap build only: SGX_TUNE_ORB_DESC_ONE_PER_WAVE */
SGX_KERNEL(64) k_orient_desc2(SgxOrbGeom g, const uint8_t *gray, int gray_pitch, const uint8_t *pyr, const uint8_t *blur,
                              const uint32_t *sel, const int *sel_count, unsigned long long umax_packed, const signed char *pattern,
                              uint8_t *kps_raw, uint8_t *desc, int *count, int cap, int batch, uint32_t *status)
{
    SGX_LDS uint32_t patch_dw[31 * SGX_MS / 4];
    SGX_LDS uint8_t bits[256];
    SGX_LDS int s_m01, s_m10;
    SGX_LDS float s_a, s_b;
    const uint8_t *patch = (const uint8_t *)patch_dw;
    int slot, frame;
    {
        const int n = (int)blockIdx.x, kc = g.kp_cap;
        if ((batch & 7) == 0) { const int xcd = n & 7, j = n >> 3; frame = xcd + 8 * (j / kc); slot = j % kc; }
        else { frame = n / kc; slot = n % kc; }
    }
    int level = -1, base = 0, total = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int n = sel_count[frame * g.nlevels + l];
        if (level < 0 && slot < total + n) { level = l; base = total; }
        total += n;
    }
    if (slot == 0) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) { count[frame] = total < cap ? total : cap; if (total > cap) sgx_atomic_or(status, SGX_ST_KP_OVERFLOW); }
        SGX_THREADS_END
    }
    if (level < 0 || slot >= cap) return;
    const SgxLevel L = g.lv[level];
    const uint32_t e = sel[((size_t)frame * g.nlevels + level) * SGX_OCT_MAXN + (slot - base)];
    const int kx = (int)(e & 0xFFF) + SGX_BORDER, ky = (int)((e >> 12) & 0xFFF) + SGX_BORDER;   // :844-845
    int stride;
    const uint8_t *img = sgx_level_ptr(g, gray, gray_pitch, pyr, frame, level, &stride);
    const int px0 = kx - 15, py0 = ky - 15, lead = px0 & 3, xa = px0 - lead;
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { s_m01 = 0; s_m10 = 0; }
#pragma unroll
    for (int it_ = 0; it_ < (31 * (SGX_MS / 4) + 63) / 64; it_++) { const int i = tid + 64 * it_; if (i >= 31 * (SGX_MS / 4)) break;
        const int r = i / (SGX_MS / 4), q = i - r * (SGX_MS / 4);
        patch_dw[i] = *(const uint32_t *)(img + (size_t)(py0 + r) * stride + xa + 4 * q);              // 36 bytes per row from an aligned start: inside the padded row
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int it_ = 0; it_ < (31 * 31 + 63) / 64; it_++) { const int i = tid + 64 * it_; if (i >= 31 * 31) break;
        const int v = i / 31 - 15, u = i - (v + 15) * 31 - 15;
        const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
        if (au <= (int)((umax_packed >> (4 * av)) & 15ull)) {
            const int I = patch[(15 + v) * SGX_MS + lead + 15 + u];
            m10 += u * I; m01 += v * I;
        }
    }
    sgx_atomic_add(&s_m10, m10); sgx_atomic_add(&s_m01, m01);
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) {
        const float angle = sgx_fast_atan2((float)s_m01, (float)s_m10);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float sn, cs;
        sgx_sincosf(angle * factorPI, &sn, &cs);
        s_a = cs; s_b = sn;
        float *kp = (float *)(kps_raw + ((size_t)frame * cap + slot) * 28);
        float fx = (float)kx, fy = (float)ky;
        if (level != 0) { fx = fx * L.scale; fy = fy * L.scale; }
        kp[0] = fx; kp[1] = fy; kp[2] = (float)L.patch_size; kp[3] = angle; kp[4] = (float)(e >> 24);
        ((int *)kp)[5] = level; ((int *)kp)[6] = -1;
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const float a = s_a, b = s_b;
    const uint8_t *bl = blur + (size_t)frame * g.blur_pitch + L.boff + (size_t)ky * L.bstride + kx;
#pragma unroll
    for (int it_ = 0; it_ < 4; it_++) { const int t = tid + 64 * it_;
        const uint32_t pw = *(const uint32_t *)(pattern + 4 * t);
        const float x0 = (float)(signed char)(pw & 255u), y0 = (float)(signed char)((pw >> 8) & 255u), x1 = (float)(signed char)((pw >> 16) & 255u), y1 = (float)(signed char)(pw >> 24);
        const int r0 = sgx_cvround(x0 * b + y0 * a), c0 = sgx_cvround(x0 * a - y0 * b);
        const int r1 = sgx_cvround(x1 * b + y1 * a), c1 = sgx_cvround(x1 * a - y1 * b);
        const int t0 = bl[r0 * L.bstride + c0], t1 = bl[r1 * L.bstride + c1];
        bits[t] = (uint8_t)(t0 < t1);
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid < 32) {
        int v = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) v |= bits[8 * tid + j] << j;
        desc[((size_t)frame * cap + slot) * 32 + tid] = (uint8_t)v;
    }
    SGX_THREADS_END
}
#endif      /* SGX_DEBUG_TAPS */

// ---------------------------------------------------------------------------------------------
// k_orient_desc4: k_orient_desc2 with FOUR keypoints per wave (16 lanes each).  The serial part of a keypoint (fastAtan2 + the bit-exact sincosf, ~250
// wave instructions executed for one active lane) is then shared by four keypoints, and the moment sums walk the staged patch as dwords (4 pixels per
// lane step).  Same arithmetic per keypoint, so identical bytes.  block -> (group of 4 slots, frame), XCD-aware like k_orient_desc.
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(64) k_orient_desc4(SgxOrbGeom g, const uint8_t *gray, int gray_pitch, const uint8_t *pyr, const uint8_t *blur,
                              const uint32_t *sel, const int *sel_count, unsigned long long umax_packed, const signed char *pattern,
                              uint8_t *kps_raw, uint8_t *desc, int *count, int cap, int batch, uint32_t *status)
{
    SGX_LDS uint32_t patch_dw[4][31 * SGX_MS / 4];
    SGX_LDS uint8_t bits[4][256];
    SGX_LDS int s_m01[4], s_m10[4];
    SGX_LDS float s_a[4], s_b[4];
    int slot4, frame;
    {
        const int n = (int)blockIdx.x, kc = (g.kp_cap + 3) >> 2;
        if ((batch & 7) == 0) { const int xcd = n & 7, j = n >> 3; frame = xcd + 8 * (j / kc); slot4 = j % kc; }
        else { frame = n / kc; slot4 = n % kc; }
    }
    // per-level counts of this frame (wave-uniform)
    int total = 0;
    for (int l = 0; l < g.nlevels; l++) total += sel_count[frame * g.nlevels + l];
    if (slot4 == 0) {
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) { count[frame] = total < cap ? total : cap; if (total > cap) sgx_atomic_or(status, SGX_ST_KP_OVERFLOW); }
        SGX_THREADS_END
    }
    if (4 * slot4 >= total || 4 * slot4 >= cap) return;

    SGX_THREADS_BEGIN(tid)
    if (tid < 4) { s_m01[tid] = 0; s_m10[tid] = 0; }
    // stage the four 31x31 moment patches (aligned dwords, 36 bytes per row)
    const int grp = tid >> 4, l16 = tid & 15, slot = 4 * slot4 + grp;
    int level = -1, base = 0, acc_ = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int n = sel_count[frame * g.nlevels + l];
        if (level < 0 && slot < acc_ + n) { level = l; base = acc_; }
        acc_ += n;
    }
    if (level >= 0 && slot < cap) {
        const uint32_t e = sel[((size_t)frame * g.nlevels + level) * SGX_OCT_MAXN + (slot - base)];
        const int kx = (int)(e & 0xFFF) + SGX_BORDER, ky = (int)((e >> 12) & 0xFFF) + SGX_BORDER;
        int stride;
        const uint8_t *img = sgx_level_ptr(g, gray, gray_pitch, pyr, frame, level, &stride);
        const int px0 = kx - 15, py0 = ky - 15, lead = px0 & 3, xa = px0 - lead;
        for (int i = l16; i < 31 * (SGX_MS / 4); i += 16) {
            const int r = i / (SGX_MS / 4), q = i - r * (SGX_MS / 4);
            patch_dw[grp][i] = *(const uint32_t *)(img + (size_t)(py0 + r) * stride + xa + 4 * q);
        }
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int grp = tid >> 4, l16 = tid & 15, slot = 4 * slot4 + grp;
    int level = -1, base = 0, acc_ = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int n = sel_count[frame * g.nlevels + l];
        if (level < 0 && slot < acc_ + n) { level = l; base = acc_; }
        acc_ += n;
    }
    if (level >= 0 && slot < cap) {
        const uint32_t e = sel[((size_t)frame * g.nlevels + level) * SGX_OCT_MAXN + (slot - base)];
        const int lead = ((int)(e & 0xFFF) + SGX_BORDER - 15) & 3;
        int m10 = 0, m01 = 0;
        for (int i = l16; i < 31 * (SGX_MS / 4); i += 16) {
            const int r = i / (SGX_MS / 4), q = i - r * (SGX_MS / 4);
            const int v = r - 15, av = v < 0 ? -v : v, um = (int)((umax_packed >> (4 * av)) & 15ull);
            const uint32_t w = patch_dw[grp][i];
            int rowsum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int u = 4 * q + j - lead - 15, au = u < 0 ? -u : u;
                const int I = au <= um ? (int)((w >> (8 * j)) & 255u) : 0;           // columns outside the disc (incl. the dword padding) weigh 0
                m10 += u * I; rowsum += I;
            }
            m01 += v * rowsum;
        }
        sgx_atomic_add(&s_m10[grp], m10); sgx_atomic_add(&s_m01[grp], m01);
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int grp = tid >> 4, l16 = tid & 15, slot = 4 * slot4 + grp;
    int level = -1, base = 0, acc_ = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int n = sel_count[frame * g.nlevels + l];
        if (level < 0 && slot < acc_ + n) { level = l; base = acc_; }
        acc_ += n;
    }
    if (level >= 0 && slot < cap && l16 == 0) {
        const SgxLevel L = g.lv[level];
        const uint32_t e = sel[((size_t)frame * g.nlevels + level) * SGX_OCT_MAXN + (slot - base)];
        const int kx = (int)(e & 0xFFF) + SGX_BORDER, ky = (int)((e >> 12) & 0xFFF) + SGX_BORDER;
        const float angle = sgx_fast_atan2((float)s_m01[grp], (float)s_m10[grp]);
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float sn, cs;
        sgx_sincosf(angle * factorPI, &sn, &cs);
        s_a[grp] = cs; s_b[grp] = sn;
        float *kp = (float *)(kps_raw + ((size_t)frame * cap + slot) * 28);
        float fx = (float)kx, fy = (float)ky;
        if (level != 0) { fx = fx * L.scale; fy = fy * L.scale; }
        kp[0] = fx; kp[1] = fy; kp[2] = (float)L.patch_size; kp[3] = angle; kp[4] = (float)(e >> 24);
        ((int *)kp)[5] = level; ((int *)kp)[6] = -1;
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int grp = tid >> 4, l16 = tid & 15, slot = 4 * slot4 + grp;
    int level = -1, base = 0, acc_ = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int n = sel_count[frame * g.nlevels + l];
        if (level < 0 && slot < acc_ + n) { level = l; base = acc_; }
        acc_ += n;
    }
    if (level >= 0 && slot < cap) {
        const SgxLevel L = g.lv[level];
        const uint32_t e = sel[((size_t)frame * g.nlevels + level) * SGX_OCT_MAXN + (slot - base)];
        const int kx = (int)(e & 0xFFF) + SGX_BORDER, ky = (int)((e >> 12) & 0xFFF) + SGX_BORDER;
        const float a = s_a[grp], b = s_b[grp];
        const uint8_t *bl = blur + (size_t)frame * g.blur_pitch + L.boff + (size_t)ky * L.bstride + kx;
#pragma unroll 4
        for (int it_ = 0; it_ < 16; it_++) { const int t = l16 + 16 * it_;
            const uint32_t pw = *(const uint32_t *)(pattern + 4 * t);
            const float x0 = (float)(signed char)(pw & 255u), y0 = (float)(signed char)((pw >> 8) & 255u), x1 = (float)(signed char)((pw >> 16) & 255u), y1 = (float)(signed char)(pw >> 24);
            const int r0 = sgx_cvround(x0 * b + y0 * a), c0 = sgx_cvround(x0 * a - y0 * b);
            const int r1 = sgx_cvround(x1 * b + y1 * a), c1 = sgx_cvround(x1 * a - y1 * b);
            const int t0 = bl[r0 * L.bstride + c0], t1 = bl[r1 * L.bstride + c1];
            bits[grp][t] = (uint8_t)(t0 < t1);
        }
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int grp = tid >> 4, l16 = tid & 15, slot = 4 * slot4 + grp;
    if (slot < total && slot < cap) {
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            const int byte = 2 * l16 + h2;
            int v = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) v |= bits[grp][8 * byte + j] << j;
            desc[((size_t)frame * cap + slot) * 32 + byte] = (uint8_t)v;
        }
    }
    SGX_THREADS_END
}
