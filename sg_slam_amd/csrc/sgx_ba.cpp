// sgx_ba.cpp — host side of the LocalBundleAdjustment C-ABI: flattening helpers (CSR by landmark / by pose), the
// Levenberg-Marquardt control loop (statement-for-statement OptimizationAlgorithmLevenberg::solve,
// G/core/optimization_algorithm_levenberg.cpp:61-164) and the kernel launches.  Reference: src/sg-slam/src/Optimizer.cc:453-778.
// fp64 solver arithmetic: multiply-adds may fuse here.  The reference does NOT fuse them — g2o and sg-slam are built plain -O3 (Thirdparty/g2o/CMakeLists.txt:57,
// src/sg-slam/CMakeLists.txt:11-12; only DBoW2 has -march=native), so on x86-64 every product is rounded before it is added — which makes this a deliberate
// divergence inside the stated tolerance: the parity bar for poses / landmarks is 1e-5 relative, not bit equality, and the fused form saves a quarter of the fp64
// instructions.  Building with -DSGX_FP_CONTRACT_OFF keeps the reference's rounding (tests/test_poseopt_gpu.py runs one parity case on that build
// when it is present).  The bit-exact integer / fp32 feature kernels keep -ffp-contract=off.
#ifndef SGX_FP_CONTRACT_OFF
#pragma clang fp contract(fast)
#endif
#include "sgx_ba_kernels.h"
#include "sgx_eg_kernels.h"
#include "sgx_prof.h"
#include "../../include/sgx.h"
#ifdef SGX_DEBUG_TAPS
#include "../../include/sgx_debug.h"      // test / tuning taps: compiled into tests/taps/libsgx_taps.so and the emulator only
#endif
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

namespace {
// One grow-only device arena per process (LocalBundleAdjustment runs on the single LocalMapping thread, LocalMapping.cc:81):
// avoids ~30 hipMalloc/hipFree pairs per call.  Layout is computed twice: first pass sizes, second pass assigns.
struct Arena {
    char *base = nullptr; size_t cap = 0, off = 0; int device = -1;
    int reserve(size_t bytes) {
#ifndef SGX_EMU
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return SGX_ERR_DEVICE;
        if (base && cur != device) {                      // the arena lives on the device that was current when it was allocated: a caller on another device gets a fresh one
            const int keep = cur; (void)hipSetDevice(device); (void)hipFree(base); (void)hipSetDevice(keep);
            base = nullptr; cap = 0;
        }
        device = cur;
#endif
        if (bytes <= cap) return SGX_OK;
        if (base) (void)hipFree(base);
        base = nullptr; cap = 0;
        const size_t want = bytes + bytes / 4;
        if (hipMalloc((void **)&base, want) != hipSuccess) return SGX_ERR_NOMEM;
        cap = want; return SGX_OK;
    }
    template <class T> void take(T **p, size_t n) { off = (off + 255) & ~(size_t)255; if (base) *p = (T *)(base + off); off += (n ? n : 1) * sizeof(T); }
};
static Arena g_arena;
// pinned staging buffer of the packed host->device inputs, kept like the arena (grow-only): no 17 MB zero-fill + pageable copy per call at 2 000 keyframes
static char *g_stage = nullptr; static size_t g_stage_cap = 0;
static char *stage_reserve(size_t bytes)
{
    if (bytes <= g_stage_cap) return g_stage;
#ifndef SGX_EMU
    if (g_stage) (void)hipHostFree(g_stage);
    g_stage = nullptr; g_stage_cap = 0;
    if (hipHostMalloc((void **)&g_stage, bytes + bytes / 4, hipHostMallocPortable) != hipSuccess) { g_stage = nullptr; return nullptr; }
#else
    free(g_stage); g_stage = (char *)malloc(bytes + bytes / 4); if (!g_stage) { g_stage_cap = 0; return nullptr; }
#endif
    g_stage_cap = bytes + bytes / 4; return g_stage;
}
static thread_local int g_ba_init_mode = 0;            // test tap (sgx_ba_debug_set_init): 0 = envelope solver: only its tiles are initialised (default), 1 = the whole matrix, 2 = the whole matrix NaN, then the tiles
static thread_local int g_ba_jobs_host = 0;            // test tap (sgx_ba_debug_set_jobs): 1 = build the Schur job list on the host (the emulator's path; A/B arm of the device builder)
static thread_local int g_ba_solver = -1;              // test / tuning tap (sgx_ba_debug_set_solver): -1 = SGX_BA_SOLVER or auto, 0 auto, 1 dense blocked Cholesky, 2 envelope solver

struct BA {
    int np, nl, ne, nf, NP;
    SgxCam cam; double dMono, dStereo;
    const volatile int32_t *stop;
    // device
    SgxBaEdge *E; SgxSE3 *T, *Tb; double *X, *Xb, *err, *Hll, *bl, *Hpl, *W, *Hpp, *bp, *S, *coef, *xp, *xl, *Dinv, *dwork, *partial;
    int *pt_start, *pt_edges, *pose_start, *pose_edges, *hidx, *free_pose, *ok; uint8_t *pt_active;
    SgxBaJob *jobs; int *blk_start; double *Linv, *xsol; long long njobs, nblk; size_t jobs_cap;
    int *pose_edges_l, *row_jobs, *row_blks, *job_off, *blk_off, *jtot;      // device-side job list build (k_ba_jobs_*): a pose's edges in ascending landmark order; per-row counts and their exclusive sums; { jobs, blocks }
    double *part_chi, *part_scale;      // device scalars block: [ok | part_scale[nblk_v] | part_chi[nblk_e]] read back with ONE copy per trial
    int nblk_e, nblk_v;
    int *env_rstart, *env_rows;          // narrow-envelope solver: rows of column step k = env_rows[env_rstart[k] .. env_rstart[k+1]) (NULL: dense solver)
    int env_nrows = 0;                   // entries of env_rows
    int env_nA, env_nB;                  // two-branch elimination: column steps [0, nA) and [nA, nA + nB) are independent, the rest is their separator (nB = 0: one branch)
    double *env_S2, *env_x2;             // the second branch's contributions to the separator (nsep x nsep, nsep)
    std::vector<double> hpart;
    // host copy of the sorted Schur job list of THIS problem (build_jobs): the classification pass filters it in place instead of rebuilding it.  Owned by the problem, so the
    // cache can neither be applied to another problem that happens to have the same job count nor outlive the call (it used to be thread_local and never released)
    std::vector<SgxBaJob> h_jobs; std::vector<int> h_blk_start;
};

static bool stopped(const BA &B) { return B.stop && *B.stop; }

static int sum_partials(BA &B, int n, double *out, bool is_max = false)
{
    B.hpart.resize(n);
    SGX_CHECK_HIP(hipMemcpy(B.hpart.data(), B.partial, sizeof(double) * n, hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < n; i++) s = is_max ? fmax(s, B.hpart[i]) : s + B.hpart[i];
    *out = s;
    return SGX_OK;
}

static int active_chi2(BA &B, double *chi)
{
    SGX_LAUNCH(k_ba_errors, dim3(B.nblk_e), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.ne, B.E, B.T, B.X, B.cam, B.dMono, B.dStereo, B.err, B.part_chi);
    B.hpart.resize(B.nblk_e);
    SGX_CHECK_HIP(hipMemcpy(B.hpart.data(), B.part_chi, sizeof(double) * B.nblk_e, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < B.nblk_e; i++) s += B.hpart[i];
    *chi = s;
    return SGX_OK;
}

// after a trial: ok flag, computeScale partials and the new chi2 partials in one device->host copy
static int read_trial(BA &B, int *ok, double *scale, double *chi)
{
    const int nd = 1 + B.nblk_v + B.nblk_e;
    B.hpart.resize(nd);
    SGX_CHECK_HIP(hipMemcpy(B.hpart.data(), (const double *)B.ok, sizeof(double) * nd, hipMemcpyDeviceToHost));
    int okv; memcpy(&okv, &B.hpart[0], 4); *ok = okv;
    double s = 0; for (int i = 0; i < B.nblk_v; i++) s += B.hpart[1 + i]; *scale = s;
    double c = 0; for (int i = 0; i < B.nblk_e; i++) c += B.hpart[1 + B.nblk_v + i]; *chi = c;
    return SGX_OK;
}

// Schur job list for the current active edge set: all ordered pairs (k1, k2) of active free-pose edges of one landmark, grouped by destination block (i1, i2) of the reduced
// system; inside a block the jobs are in landmark order, the order in which the reference subtracts them (block_solver.hpp:380-433); k_ba_schur_pairs then sums every block
// sequentially, without atomics.  Built block row by block row: the jobs of row i1 come from the edges of that pose (pose_edges_l: ascending landmark) x the active edges of
// their landmarks, and are dealt to their i2 with a counting sort over the few poses the row touches — everything a row needs sits in L1 / L2 (a sort of the whole list by
// the nf^2 keys was 30 ms per call at 2 000 keyframes: a third of the bundle adjustment).
// level1 == nullptr: build from scratch (every edge of a free pose is active).  level1 != nullptr: the list of the SAME problem after the classification pass — the previous list
// with the jobs of the newly switched-off edges taken out (a stable filter of the sorted list: a few ms instead of a rebuild).
static int build_jobs(BA &B, const std::vector<int> &pt_start, const std::vector<int> &pt_edges, const std::vector<int> &pose_start, const std::vector<int> &pose_edges_l,
                      const std::vector<SgxBaEdge> &E, const std::vector<uint8_t> *level1, const std::vector<int> &hidx, const std::vector<int> &free_pose)
{
    std::vector<SgxBaJob> &sorted = B.h_jobs;
    std::vector<int> &blk_start = B.h_blk_start;
    if (level1 && !sorted.empty() && (long long)sorted.size() == B.njobs) {
        const std::vector<uint8_t> &lv = *level1;
        size_t w = 0, nb = 0;
        const size_t nblk = blk_start.size() - 1;
        for (size_t b = 0; b < nblk; b++) {
            const size_t beg = (size_t)blk_start[b], end = (size_t)blk_start[b + 1], w0 = w;
            for (size_t i = beg; i < end; i++) { const SgxBaJob j = sorted[i]; if (!lv[(size_t)j.k1] && !lv[(size_t)j.k2]) sorted[w++] = j; }
            if (w > w0) blk_start[nb++] = (int)w0;
        }
        sorted.resize(w); blk_start.resize(nb + 1); blk_start[nb] = (int)w;
    } else {
        const int nthreads = std::max(1, std::min(8, B.nf / 64));
        std::vector<int> eh(B.ne);
        for (int k = 0; k < B.ne; k++) eh[k] = (level1 && (*level1)[k]) ? -1 : hidx[E[k].pose];      // destination row / column of an active edge, -1 = not in the system
        // block rows dealt to the threads in contiguous ranges of about equal edge counts
        std::vector<int> cut(nthreads + 1, B.nf); cut[0] = 0;
        { long long tot = 0; for (int h = 0; h < B.nf; h++) tot += pose_start[free_pose[h] + 1] - pose_start[free_pose[h]];
          long long run = 0; int t = 1; for (int h = 0; h < B.nf && t < nthreads; h++) { run += pose_start[free_pose[h] + 1] - pose_start[free_pose[h]]; if (run * nthreads >= tot * t) cut[t++] = h + 1; } }
        std::vector<std::vector<SgxBaJob>> part((size_t)nthreads); std::vector<std::vector<int>> pblk((size_t)nthreads);
        auto work = [&](int t) {
            std::vector<SgxBaJob> &out = part[(size_t)t], grp; std::vector<int> &ob = pblk[(size_t)t], grp_h2, touched, cnt((size_t)(B.nf > 0 ? B.nf : 1), 0);
            for (int h1 = cut[t]; h1 < cut[t + 1]; h1++) {
                const int p = free_pose[h1];
                grp.clear(); grp_h2.clear(); touched.clear();
                for (int q = pose_start[p]; q < pose_start[p + 1]; q++) {
                    const int k1 = pose_edges_l[q]; if (eh[k1] < 0) continue;
                    const int l = E[k1].point;
                    for (int q2 = pt_start[l]; q2 < pt_start[l + 1]; q2++) {
                        const int k2 = pt_edges[q2], h2 = eh[k2]; if (h2 < 0) continue;
                        if (cnt[h2]++ == 0) touched.push_back(h2);
                        grp.push_back(SgxBaJob{k1, k2}); grp_h2.push_back(h2);
                    }
                }
                if (grp.empty()) continue;
                std::sort(touched.begin(), touched.end());
                const size_t base = out.size();
                { int run = 0; for (int h2 : touched) { const int c = cnt[h2]; cnt[h2] = run; ob.push_back((int)base + run); run += c; } }
                out.resize(base + grp.size());
                for (size_t i = 0; i < grp.size(); i++) out[base + (size_t)cnt[grp_h2[i]]++] = grp[i];
                for (int h2 : touched) cnt[h2] = 0;
            }
        };
        if (nthreads == 1) work(0);
        else { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
        size_t total = 0; for (auto &v : part) total += v.size();
        if (total > B.jobs_cap) return SGX_ERR_NOMEM;
        sorted.resize(total); blk_start.clear();
        size_t off = 0;
        for (int t = 0; t < nthreads; t++) {
            if (!part[(size_t)t].empty()) memcpy(sorted.data() + off, part[(size_t)t].data(), sizeof(SgxBaJob) * part[(size_t)t].size());
            for (int b : pblk[(size_t)t]) blk_start.push_back((int)off + b);
            off += part[(size_t)t].size();
        }
        blk_start.push_back((int)total);
    }
    B.njobs = (long long)sorted.size(); B.nblk = 0;
    if (sorted.empty()) return SGX_OK;
    B.nblk = (long long)blk_start.size() - 1;
    SGX_CHECK_HIP(hipMemcpy(B.jobs, sorted.data(), sizeof(SgxBaJob) * sorted.size(), hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(B.blk_start, blk_start.data(), sizeof(int) * blk_start.size(), hipMemcpyHostToDevice));
    return SGX_OK;
}

// The same list built on the device (k_ba_jobs_row / k_ba_jobs_scan, sgx_ba_kernels.h): no host pass over the edges, no 25 MB upload at 2 000 keyframes, and after the
// classification pass no download of the edge records either — the kernels read the levels where k_ba_classify wrote them.  One 8-byte read-back (jobs, blocks) per build.
static int build_jobs_dev(BA &B)
{
#ifndef SGX_EMU
    B.njobs = 0; B.nblk = 0;
    if (B.nf < 1) return SGX_OK;
    const size_t lds = sizeof(int) * (size_t)B.nf;
    hipLaunchKernelGGL((k_ba_jobs_row<0>), dim3((unsigned)B.nf), dim3(64), lds, (sgx_stream_t)0, B.nf, B.free_pose, B.pose_start, B.pose_edges_l, B.pt_start, B.pt_edges, B.E, B.hidx,
                       B.row_jobs, B.row_blks, (const int *)nullptr, (const int *)nullptr, (SgxBaJob *)nullptr, (int *)nullptr);
    SGX_LAUNCH(k_ba_jobs_scan, dim3(1), dim3(256), (sgx_stream_t)0, B.nf, B.row_jobs, B.row_blks, B.job_off, B.blk_off, B.jtot, B.blk_start);
    int tot[2] = { 0, 0 };
    SGX_CHECK_HIP(hipMemcpy(tot, B.jtot, sizeof tot, hipMemcpyDeviceToHost));
    if ((size_t)tot[0] > B.jobs_cap) return SGX_ERR_NOMEM;
    B.njobs = tot[0]; B.nblk = tot[1];
    if (tot[0] == 0) return SGX_OK;
    hipLaunchKernelGGL((k_ba_jobs_row<1>), dim3((unsigned)B.nf), dim3(64), lds, (sgx_stream_t)0, B.nf, B.free_pose, B.pose_start, B.pose_edges_l, B.pt_start, B.pt_edges, B.E, B.hidx,
                       B.row_jobs, B.row_blks, B.job_off, B.blk_off, B.jobs, B.blk_start);
    SGX_CHECK_HIP(hipGetLastError());
    return SGX_OK;
#else
    (void)B; return SGX_ERR_UNSUPPORTED;
#endif
}

// Dense symmetric positive definite solve  S x = bp - coef  on the device (in place: S is overwritten by its factor): register / LDS kernels for small systems, the blocked
// right-looking Cholesky with the fp64-MFMA trailing update above SGX_CHOL_SMALL unknowns.  *ok (device) is cleared when a pivot is not positive; x then keeps its previous
// content.  *xout = where the solution was left (xp or xsol).  Shared by the bundle adjustments (reduced camera system) and the essential-graph optimisation.
struct Chol { int NP; double *S, *Linv, *bp, *coef, *xp, *xsol; int *ok; const int *env_rstart = nullptr, *env_rows = nullptr; int env_nA = 0, env_nB = 0; double *env_S2 = nullptr, *env_x2 = nullptr; };      // env_*: column-step row lists of a narrow envelope (NULL = dense)
static int chol_factor_solve(const Chol &C, const double **xout)
{
    *xout = C.xp;
    if (C.env_rstart && C.NP > 0) {                      // sparse covisibility: the whole factorisation + forward substitution as one persistent workgroup, then the backward pass
        const int nt = (C.NP + SGX_NB - 1) / SGX_NB;
        static const int env_dbg = sgx_getenv("SGX_ENV_DBG") ? atoi(sgx_getenv("SGX_ENV_DBG")) : 0;      // timing tap: 1 skip the diagonal tiles, 2 skip panel + update, 4 skip the update
        const int nA = C.env_nB > 0 ? C.env_nA : nt, nB = C.env_nB > 0 ? C.env_nB : 0;
        if (nB > 0) {                                    // the second branch's separator contributions start from zero
            const size_t ns = (size_t)(C.NP - (nA + nB) * SGX_NB);
            SGX_CHECK_HIP(hipMemsetAsync(C.env_S2, 0, sizeof(double) * (ns * ns + ns), 0));      // x2 follows S2 in the arena
        }
        SGX_LAUNCH(k_chol_env_factor, dim3(nB > 0 ? 2 : 1), dim3(SGX_ENV_THREADS), (sgx_stream_t)0, C.NP, nt, C.env_rstart, C.env_rows, C.S, C.Linv, C.ok, C.bp, C.coef, C.xp, env_dbg, 0, nA, nB, C.env_S2, C.env_x2);
        if (nA + nB < nt) {
            SGX_LAUNCH(k_chol_env_factor, dim3(1), dim3(SGX_ENV_THREADS), (sgx_stream_t)0, C.NP, nt, C.env_rstart, C.env_rows, C.S, C.Linv, C.ok, C.bp, C.coef, C.xp, env_dbg, 1, nA, nB, C.env_S2, C.env_x2);
            SGX_LAUNCH(k_chol_env_back, dim3(1), dim3(1024), (sgx_stream_t)0, C.NP, nt, C.env_rstart, C.env_rows, C.S, C.Linv, C.xp, C.xsol, C.ok, 0, nA, nB);
        }
        SGX_LAUNCH(k_chol_env_back, dim3(nB > 0 ? 2 : 1), dim3(1024), (sgx_stream_t)0, C.NP, nt, C.env_rstart, C.env_rows, C.S, C.Linv, C.xp, C.xsol, C.ok, 1, nA, nB);
        *xout = C.xsol;
        return SGX_OK;
    }
    // workgroup sizes of the single-workgroup solver kernels (env = tuning taps): their phases are short, so fewer waves mean cheaper barriers
    static const int t_small = sgx_getenv("SGX_TUNE_CHOL_SMALL_THREADS") ? atoi(sgx_getenv("SGX_TUNE_CHOL_SMALL_THREADS")) : 256;
    static const int t_diag = sgx_getenv("SGX_TUNE_CHOL_DIAG_THREADS") ? atoi(sgx_getenv("SGX_TUNE_CHOL_DIAG_THREADS")) : 256;
    static const int t_solve = sgx_getenv("SGX_TUNE_CHOL_SOLVE_THREADS") ? atoi(sgx_getenv("SGX_TUNE_CHOL_SOLVE_THREADS")) : 256;
    if (C.NP > 0 && C.NP <= SGX_CHOL_SMALL) {
        static const int small_lds = sgx_getenv("SGX_TUNE_CHOL_SMALL_LDS") ? atoi(sgx_getenv("SGX_TUNE_CHOL_SMALL_LDS")) : 0;     // 1 = the LDS-resident version (comparison tap)
        if (small_lds) SGX_LAUNCH(k_chol_small, dim3(1), dim3(t_small), (sgx_stream_t)0, C.NP, C.S, C.bp, C.coef, C.xp, C.ok);
        else SGX_LAUNCH(k_chol_small_reg, dim3(1), dim3(256), (sgx_stream_t)0, C.NP, C.S, C.bp, C.coef, C.xp, C.ok);
    } else if (C.NP > 0) {                                   // blocked Cholesky of the reduced camera system
        const int nt = (C.NP + SGX_NB - 1) / SGX_NB;
        // tiles per outer panel; small systems keep one level (a rank-256 launch on the critical path costs them more than its eight rank-32 shares)
        static const int wide_min = sgx_getenv("SGX_TUNE_CHOL_WIDE_MIN") ? atoi(sgx_getenv("SGX_TUNE_CHOL_WIDE_MIN")) : 1024;
        const int OT = C.NP > wide_min ? SGX_OB / SGX_NB : (1 << 24);
        for (int kb = 0; kb < nt; kb++) {
            const int k0 = kb * SGX_NB, rem = nt - kb - 1;
            const int in_panel = OT - 1 - kb % OT;           // column tiles right of this one that still belong to the outer panel
#ifndef SGX_EMU
            static const int diag_lds = sgx_getenv("SGX_TUNE_CHOL_DIAG_LDS") ? atoi(sgx_getenv("SGX_TUNE_CHOL_DIAG_LDS")) : 0;      // 1 = the workgroup / LDS version (comparison tap)
            if (!diag_lds) SGX_LAUNCH(k_chol_diag_wave, dim3(1), dim3(64), (sgx_stream_t)0, C.NP, k0, C.S, C.Linv, C.ok, C.bp, C.coef, C.xp);
            else
#endif
            SGX_LAUNCH(k_chol_diag, dim3(1), dim3(t_diag), (sgx_stream_t)0, C.NP, k0, C.S, C.Linv, C.ok, C.bp, C.coef, C.xp);
            if (rem > 0) {
                SGX_LAUNCH(k_chol_panel, dim3(rem), dim3(256), (sgx_stream_t)0, C.NP, k0, C.S, C.Linv, C.ok, C.xp);
                const int pc = in_panel < rem ? in_panel : rem;
                if (pc > 0) SGX_LAUNCH(k_chol_update, dim3(rem, pc), dim3(256), (sgx_stream_t)0, C.NP, k0, C.S, C.ok);
                if (in_panel == 0) {                         // panel finished: one rank-256 update of the rest on the matrix cores
                    const int p0 = (kb / OT) * SGX_OB, q0 = p0 + SGX_OB;
                    const int wt = (C.NP - q0 + SGX_WT - 1) / SGX_WT;
                    if (wt > 0) SGX_LAUNCH(k_chol_update_wide, dim3(wt, wt), dim3(256), (sgx_stream_t)0, C.NP, p0, SGX_OB, C.S, C.ok);
                }
            }
        }
        static const int back_min = sgx_getenv("SGX_TUNE_CHOL_BACK_MIN") ? atoi(sgx_getenv("SGX_TUNE_CHOL_BACK_MIN")) : 0;     // measured: the per-block launches win at every blocked size (360 unknowns: 9.9 -> 9.5 ms per LocalBA, 12 000: 2.5 -> 1.1 s)
        if (C.NP < back_min) {
            SGX_LAUNCH(k_chol_solve, dim3(1), dim3(t_solve), (sgx_stream_t)0, C.NP, C.S, C.Linv, C.bp, C.coef, C.xp, C.ok);
        } else {                                            // one launch per diagonal block, all CUs on the row panel (k_chol_back_step)
            for (int kb = nt - 1; kb >= 0; kb--) {
                const int k0 = kb * SGX_NB;
                SGX_LAUNCH(k_chol_back_step, dim3(k0 > 0 ? (k0 + 255) / 256 : 1), dim3(256), (sgx_stream_t)0, C.NP, k0, C.S, C.Linv, C.xp, C.xsol, C.ok);
            }
            *xout = C.xsol;
        }
    }
    return SGX_OK;
}

// one optimizer.optimize(iterations) call
static int optimize(BA &B, int iterations, int *iters_done, double *final_chi)
{
    double lambda = -1, ni = 2; int nBadLM = 0; *iters_done = 0;
    for (int it = 0; it < iterations; it++) {
        if (stopped(B)) break;
        double currentChi = 0; int rc = active_chi2(B, &currentChi); if (rc != SGX_OK) return rc;
        double tempChi = currentChi; const double iniChi = currentChi;
        // buildSystem: J^T J / J^T r block accumulation
        sgx_prof_begin(SGX_K_BA_LINEARIZE, (sgx_stream_t)0);
        SGX_LAUNCH(k_ba_linearize_points, dim3((B.nl + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nl, B.pt_start, B.pt_edges,
                   B.E, B.T, B.X, B.hidx, B.err, B.cam, B.dMono, B.dStereo, B.Hll, B.bl, B.Hpl, B.pt_active);
        if (B.nf > 0)
            SGX_LAUNCH(k_ba_linearize_poses, dim3(B.nf), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.np, B.free_pose, B.pose_start, B.pose_edges, B.E, B.T, B.X, B.err,
                       B.cam, B.dMono, B.dStereo, B.Hpp, B.bp);
        sgx_prof_end(SGX_K_BA_LINEARIZE, (sgx_stream_t)0);
        if (it == 0) {
            SGX_LAUNCH(k_ba_maxdiag, dim3(B.nblk_v), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nf, B.nl, B.Hpp, B.Hll, B.pt_active, B.partial);
            double maxd = 0; rc = sum_partials(B, B.nblk_v, &maxd, true); if (rc != SGX_OK) return rc;
            lambda = 1e-5 * maxd; ni = 2; nBadLM = 0;
        }
        double rho = 0; int qmax = 0;
        do {
            int ok2 = 1;
            const double *xsol = B.xp;                               // where the solver leaves the pose increments
            { const int one = 1; SGX_CHECK_HIP(hipMemcpyAsync(B.ok, &one, 4, hipMemcpyHostToDevice, 0)); }
            sgx_prof_begin(SGX_K_BA_SCHUR, (sgx_stream_t)0);
            if (B.NP > 0) {
                const int g = (int)(((size_t)B.NP * B.NP + SGX_BA_THREADS - 1) / SGX_BA_THREADS);
                if (B.env_rstart && g_ba_init_mode != 1) {       // envelope solver: only the tiles it reads
                    if (g_ba_init_mode == 2) SGX_CHECK_HIP(hipMemsetAsync(B.S, 0xFF, sizeof(double) * (size_t)B.NP * B.NP, 0));      // (test tap) all ones = NaN everywhere first
                    const int nt = (B.NP + SGX_NB - 1) / SGX_NB;
                    SGX_LAUNCH(k_ba_schur_init_env, dim3((unsigned)(nt + B.env_nrows)), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nf, B.Hpp, lambda, B.S, B.coef, nt, B.env_rstart, B.env_rows);
                } else
                SGX_LAUNCH(k_ba_schur_init, dim3(g > 4096 ? 4096 : g), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nf, B.Hpp, lambda, B.S, B.coef);
            }
            SGX_LAUNCH(k_ba_dinv, dim3((B.nl + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nl, B.pt_active, B.Hll, lambda, B.Dinv);
            if (B.njobs > 0) {
                const long long n36 = B.nblk * 36, n6 = (long long)B.ne * 6;
                SGX_LAUNCH(k_ba_hpl_dinv, dim3((unsigned)((n6 + SGX_BA_THREADS - 1) / SGX_BA_THREADS)), dim3(SGX_BA_THREADS), (sgx_stream_t)0, n6, B.E, B.hidx, B.Hpl, B.Dinv, B.W);
                SGX_LAUNCH(k_ba_schur_pairs, dim3((unsigned)((n36 + SGX_BA_THREADS - 1) / SGX_BA_THREADS)), dim3(SGX_BA_THREADS), (sgx_stream_t)0, n36, B.nf, B.blk_start, B.jobs, B.E,
                           B.hidx, B.bl, B.Hpl, B.W, B.S, B.coef);
            }
            sgx_prof_end(SGX_K_BA_SCHUR, (sgx_stream_t)0);
            sgx_prof_begin(SGX_K_BA_SOLVE, (sgx_stream_t)0);
            { Chol C = { B.NP, B.S, B.Linv, B.bp, B.coef, B.xp, B.xsol, B.ok }; C.env_rstart = B.env_rstart; C.env_rows = B.env_rows; C.env_nA = B.env_nA; C.env_nB = B.env_nB; C.env_S2 = B.env_S2; C.env_x2 = B.env_x2; if ((rc = chol_factor_solve(C, &xsol)) != SGX_OK) return rc; }
            sgx_prof_end(SGX_K_BA_SOLVE, (sgx_stream_t)0);
            sgx_prof_begin(SGX_K_BA_UPDATE, (sgx_stream_t)0);
            // when the factorisation failed, xp/xl keep the previous solution (as g2o's _x does) and the step is rejected below
            SGX_LAUNCH(k_ba_backsub, dim3((B.nl + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nl, B.pt_start, B.pt_edges, B.E, B.hidx,
                           B.pt_active, B.bl, B.Hpl, B.Dinv, xsol, B.xl, B.ok);
            // push + update + computeScale
            SGX_LAUNCH(k_ba_update, dim3(B.nblk_v), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.np, B.nl, B.hidx, B.pt_active, xsol, B.xl, B.bp, B.bl, lambda,
                       B.T, B.X, B.Tb, B.Xb, B.part_scale);
            SGX_LAUNCH(k_ba_errors, dim3(B.nblk_e), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.ne, B.E, B.T, B.X, B.cam, B.dMono, B.dStereo, B.err, B.part_chi);
            sgx_prof_end(SGX_K_BA_UPDATE, (sgx_stream_t)0);
            double scale = 0; rc = read_trial(B, &ok2, &scale, &tempChi); if (rc != SGX_OK) return rc;
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            scale += 1e-3; rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                const double r21 = 2 * rho - 1;
                double alpha = 1. - r21 * r21 * r21; alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                lambda *= (alpha > 1. / 3. ? alpha : 1. / 3.); ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;                               // pop
                SGX_CHECK_HIP(hipMemcpy(B.T, B.Tb, sizeof(SgxSE3) * B.np, hipMemcpyDeviceToDevice));
                SGX_CHECK_HIP(hipMemcpy(B.X, B.Xb, sizeof(double) * 3 * (size_t)B.nl, hipMemcpyDeviceToDevice));
            }
            qmax++;
        } while (rho < 0 && qmax < 10 && !stopped(B));
        *iters_done = it + 1; *final_chi = currentChi;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) break;
    }
    return SGX_OK;
}
}  // namespace

static thread_local int g_ba_last_plan[4] = { 0, 0, 0, 0 };
SGX_TAP int sgx_ba_debug_last_plan(int32_t plan[4]) { if (!plan) return SGX_ERR_INVALID; for (int i = 0; i < 4; i++) plan[i] = g_ba_last_plan[i]; return SGX_OK; }
SGX_TAP int sgx_ba_debug_set_init(int mode) { g_ba_init_mode = mode < 0 ? 0 : (mode > 2 ? 2 : mode); return SGX_OK; }
SGX_TAP int sgx_ba_debug_set_jobs(int host) { g_ba_jobs_host = host ? 1 : 0; return SGX_OK; }
SGX_TAP int sgx_ba_debug_set_solver(int mode) { g_ba_solver = mode < 0 ? -1 : (mode > 2 ? 2 : mode); return SGX_OK; }

// mode 0: Optimizer::LocalBundleAdjustment (Optimizer.cc:453-778); mode 1: Optimizer::BundleAdjustment (Optimizer.cc:49-237): one optimize(n_iterations) over
// all edges, Huber deltas sqrt(5.99) / sqrt(7.815) only when `robust`, no classification, every pose rewritten
static int ba_run(const sgx_ba_problem *P, const sgx_camera *cam, const volatile int32_t *stop_flag, uint8_t *edge_erase, sgx_ba_stats *stats, int mode, int n_iterations, int robust)
{
    if (!P || !cam || (mode == 0 && !edge_erase) || P->n_poses < 1 || P->n_points < 1 || P->n_edges < 1 || !P->poses || !P->pose_fixed || !P->points ||
        !P->edge_pose || !P->edge_point || !P->edge_obs || !P->edge_info) return SGX_ERR_INVALID;
    // the device arena is one per process: calls from several threads (the reference has one LocalMapping thread, plus GlobalBundleAdjustment from LoopClosing) take turns
    static std::mutex arena_mutex;
    std::lock_guard<std::mutex> arena_lock(arena_mutex);
    static const bool timing = sgx_getenv("SGX_BA_TIMING") != nullptr;          // tuning tap: wall-clock of the host phases on stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tmark = now();
    auto lap = [&](const char *what) { if (timing) { (void)hipDeviceSynchronize(); const double t = now(); fprintf(stderr, "[sgx_ba] %-22s %8.3f ms\n", what, t - tmark); tmark = t; } };
    BA B; memset((void *)&B, 0, offsetof(BA, hpart));
    B.np = P->n_poses; B.nl = P->n_points; B.ne = P->n_edges; B.stop = stop_flag;
    B.cam.fx = cam->fx; B.cam.fy = cam->fy; B.cam.cx = cam->cx; B.cam.cy = cam->cy; B.cam.bf = cam->bf;
    B.dMono = (double)(float)sqrt(5.991); B.dStereo = (double)(float)sqrt(7.815);          // Optimizer.cc:569-570 (float)
    if (mode == 1) { B.dMono = robust ? (double)(float)sqrt(5.99) : 1e150; B.dStereo = robust ? (double)(float)sqrt(7.815) : 1e150; }   // :83-84; no kernel = Huber that never leaves its quadratic zone
    if (stats) memset(stats, 0, sizeof *stats);
    if (edge_erase) memset(edge_erase, 0, B.ne);
    if (mode == 0 && stop_flag && *stop_flag) return SGX_OK;                                 // Optimizer.cc:655-657
    // ---- index structures
    std::vector<int> hidx(B.np), free_pose;
    for (int i = 0; i < B.np; i++) { if (P->pose_fixed[i]) hidx[i] = -1; else { hidx[i] = (int)free_pose.size(); free_pose.push_back(i); } }
    B.nf = (int)free_pose.size(); B.NP = 6 * B.nf;
    if (B.NP > SGX_BA_MAX_DENSE) return SGX_ERR_UNSUPPORTED;
    std::vector<SgxBaEdge> E(B.ne);
    std::vector<int> pt_start(B.nl + 1, 0), pose_start(B.np + 1, 0), pt_edges(B.ne), pose_edges(B.ne);
    for (int k = 0; k < B.ne; k++) {
        const int p = P->edge_pose[k], l = P->edge_point[k];
        if (p < 0 || p >= B.np || l < 0 || l >= B.nl) return SGX_ERR_INVALID;
        E[k].pose = p; E[k].point = l; E[k].flags = (P->edge_obs[3 * k + 2] < 0 ? 0 : 1) | 4;
        E[k].obs[0] = P->edge_obs[3 * k]; E[k].obs[1] = P->edge_obs[3 * k + 1]; E[k].obs[2] = P->edge_obs[3 * k + 2]; E[k].info = P->edge_info[k];
        pt_start[l + 1]++; pose_start[p + 1]++;
    }
    for (int l = 0; l < B.nl; l++) pt_start[l + 1] += pt_start[l];
    for (int p = 0; p < B.np; p++) pose_start[p + 1] += pose_start[p];
    { std::vector<int> f1(B.nl, 0), f2(B.np, 0);
      for (int k = 0; k < B.ne; k++) { pt_edges[pt_start[E[k].point] + f1[E[k].point]++] = k; pose_edges[pose_start[E[k].pose] + f2[E[k].pose]++] = k; } }
    std::vector<int> pose_edges_l(pose_edges);                                     // the edges of a pose in ascending landmark order (ties: edge order) for the Schur job list
    for (int p = 0; p < B.np; p++) {      // (a caller that adds its edges landmark by landmark — Optimizer.cc:573-640 does — hands every pose its edges already in this order)
        const auto lt = [&](int x, int y) { return E[x].point != E[y].point ? E[x].point < E[y].point : x < y; };
        if (!std::is_sorted(pose_edges_l.begin() + pose_start[p], pose_edges_l.begin() + pose_start[p + 1], lt)) std::sort(pose_edges_l.begin() + pose_start[p], pose_edges_l.begin() + pose_start[p + 1], lt);
    }
    std::vector<double> Xd(3 * (size_t)B.nl);
    for (size_t i = 0; i < Xd.size(); i++) Xd[i] = (double)P->points[i];
    // upper bound of the Schur job list: sum over landmarks of (edges with a free pose)^2
    size_t jobs_cap = 0;
    for (int l = 0; l < B.nl; l++) { size_t c = 0; for (int q = pt_start[l]; q < pt_start[l + 1]; q++) if (hidx[E[pt_edges[q]].pose] >= 0) c++; jobs_cap += c * c; }
    B.jobs_cap = jobs_cap;
    // ---- sparsity of the reduced camera system: free poses i1, i2 are coupled when they share a landmark.  With the free poses in keyframe order (hidx) the tile rows of
    // the system are non-zero from the first tile column ft[r] on and fill stays inside that envelope; when it is narrow the solver walks it with one persistent workgroup
    // (k_chol_env_factor) instead of the dense blocked factorisation.  SGX_BA_SOLVER = dense | env | auto (default).
    std::vector<int> env_rstart, env_rows;
    int env_nA = 0, env_nB = 0; size_t env_nsep = 0;
    {
        static const char *solver_env = sgx_getenv("SGX_BA_SOLVER");
        const int mode_env = !solver_env ? 0 : (strcmp(solver_env, "dense") == 0 ? 1 : (strcmp(solver_env, "env") == 0 ? 2 : 0));
        const int smode = g_ba_solver >= 0 ? g_ba_solver : mode_env;
        const bool want_env = smode != 1, force_env = smode == 2;
        const int nt = (B.NP + SGX_NB - 1) / SGX_NB;
        if (want_env && B.NP > (force_env ? 0 : 1024)) {
            // free poses of every landmark (natural order = keyframe order), and the lowest pose each pose is coupled with
            std::vector<int> fblk(B.nf);
            for (int i = 0; i < B.nf; i++) fblk[i] = i;
            for (int l = 0; l < B.nl; l++) {
                int lo = B.nf;
                for (int q = pt_start[l]; q < pt_start[l + 1]; q++) { const int h = hidx[E[pt_edges[q]].pose]; if (h >= 0 && h < lo) lo = h; }
                for (int q = pt_start[l]; q < pt_start[l + 1]; q++) { const int h = hidx[E[pt_edges[q]].pose]; if (h >= 0 && lo < fblk[h]) fblk[h] = lo; }
            }
            // Plan for an ordering pos[natural free-pose index] -> position: tile pattern of the reduced system from the landmarks' pose sets, symbolic tile Cholesky
            // (the structure of column k is R(k); every pair of R(k) becomes a tile of the factor), narrowness test.  Returns false when a step has too many rows.
            auto build = [&](const std::vector<int> &pos, std::vector<int> &rstart, std::vector<int> &rws) -> bool {
                std::vector<uint8_t> pat((size_t)nt * nt, 0);
                std::vector<int> tl;
                for (int l = 0; l < B.nl; l++) {
                    tl.clear();
                    for (int q = pt_start[l]; q < pt_start[l + 1]; q++) {
                        const int h = hidx[E[pt_edges[q]].pose]; if (h < 0) continue;
                        const int u0 = 6 * pos[h], t0 = u0 / SGX_NB, t1 = (u0 + 5) / SGX_NB;
                        tl.push_back(t0); if (t1 != t0) tl.push_back(t1);
                    }
                    std::sort(tl.begin(), tl.end()); tl.erase(std::unique(tl.begin(), tl.end()), tl.end());      // a landmark's poses sit on a few tiles
                    for (size_t a = 0; a < tl.size(); a++) for (size_t b = 0; b < a; b++) pat[(size_t)tl[a] * nt + tl[b]] = 1;
                }
                // a pose that straddles two tiles couples them even without a landmark
                for (int h = 0; h < B.nf; h++) { const int u0 = 6 * h, t0 = u0 / SGX_NB, t1 = (u0 + 5) / SGX_NB; if (t1 != t0) pat[(size_t)t1 * nt + t0] = 1; }
                rstart.assign(nt + 1, 0); rws.clear();
                std::vector<int> R;
                size_t total = 0;
                for (int k = 0; k < nt; k++) {
                    R.clear();
                    for (int r = k + 1; r < nt; r++) if (pat[(size_t)r * nt + k]) R.push_back(r);
                    if ((int)R.size() > SGX_ENV_MAXM) return false;
                    for (size_t a = 0; a < R.size(); a++) for (size_t b = 0; b < a; b++) pat[(size_t)R[a] * nt + R[b]] = 1;
                    rws.insert(rws.end(), R.begin(), R.end());                                 // rows ascending inside a step
                    rstart[k + 1] = (int)rws.size(); total += R.size();
                }
                if (rws.empty()) rws.push_back(0);
                // narrow = a step's tile products fit a few rounds of the persistent workgroup's waves; otherwise the dense two-level path (matrix cores) wins
                return force_env || total <= (size_t)nt * 10;
            };
            // Two-branch ordering: [poses 0 .. a) ascending][poses t0-1 .. bs DEScending][separator: the rest, natural order]: the band is eliminated from both ends at once.
            // The separator must cut every coupling between the halves (no pose of [bs, t0) shares a landmark with a pose < a): it is the stretch [a, bs) behind the first
            // half plus — when the trajectory closes on itself — the tail [t0, nf) that sees the start again.  Branch sizes are multiples of 16 poses = 3 tiles.
            const int twist_env = sgx_getenv("SGX_BA_TWIST") ? atoi(sgx_getenv("SGX_BA_TWIST")) : 1;          // read per call (tests switch it)
            bool done = false;
            if (twist_env && nt >= 24) {
                const int a = (B.nf / 2 / 16) * 16;
                int bs = a; while (bs < B.nf && fblk[bs] < a) bs++;                     // behind the first half: coupled with it
                int t0 = bs; while (t0 < B.nf && fblk[t0] >= a) t0++;                   // the independent stretch ends where the start is seen again
                const int nb_poses = ((t0 - bs) / 16) * 16; bs = t0 - nb_poses;
                if (a >= 16 && nb_poses >= 16) {
                    std::vector<int> pos(B.nf);
                    int sp = a + nb_poses;
                    for (int h = 0; h < B.nf; h++) pos[h] = h < a ? h : ((h >= bs && h < t0) ? a + (t0 - 1 - h) : sp++);
                    const int tA = 6 * a / SGX_NB, tB = 6 * nb_poses / SGX_NB;
                    bool ok2 = build(pos, env_rstart, env_rows);
                    // the two branches must not touch each other's tiles: no row of the second branch in a column step of the first (they run concurrently)
                    for (int k = 0; ok2 && k < tA; k++) for (int q = env_rstart[k]; q < env_rstart[k + 1]; q++) if (env_rows[q] >= tA && env_rows[q] < tA + tB) { ok2 = false; break; }
                    if (ok2) {
                        done = true; env_nA = tA; env_nB = tB; env_nsep = (size_t)B.NP - (size_t)(env_nA + env_nB) * SGX_NB;
                        // the order of the unknowns IS the order of the free poses: renumber them
                        std::vector<int> fp2(B.nf);
                        for (int h = 0; h < B.nf; h++) fp2[pos[h]] = free_pose[h];
                        free_pose.swap(fp2);
                        for (int h = 0; h < B.nf; h++) hidx[free_pose[h]] = h;
                    }
                }
            }
            if (!done) {
                std::vector<int> pos(B.nf); for (int h = 0; h < B.nf; h++) pos[h] = h;
                if (!build(pos, env_rstart, env_rows)) { env_rstart.clear(); env_rows.clear(); }
            }
        }
    }
    lap("index + solver plan");
    // ---- device state: one arena; the host->device inputs are packed contiguously and uploaded with one copy
    float *dTcw = nullptr; uint8_t *dfixed = nullptr, *derase = nullptr;
    const int nv = B.np > B.nl ? B.np : B.nl;
    B.nblk_e = (B.ne + SGX_BA_THREADS - 1) / SGX_BA_THREADS; B.nblk_v = (nv + SGX_BA_THREADS - 1) / SGX_BA_THREADS;
    int rc = SGX_OK;
    size_t in_bytes = 0;
    for (int pass = 0; pass < 2; pass++) {
        Arena &A = g_arena; A.off = 0;
        char *save = A.base; if (pass == 0) A.base = nullptr;             // sizing pass: no pointers handed out
        A.take(&B.E, B.ne); A.take(&B.X, 3 * (size_t)B.nl); A.take(&B.pt_start, B.nl + 1); A.take(&B.pt_edges, B.ne);
        A.take(&B.pose_start, B.np + 1); A.take(&B.pose_edges, B.ne); A.take(&B.hidx, B.np); A.take(&B.free_pose, B.nf);
        A.take(&dTcw, 16 * (size_t)B.np); A.take(&dfixed, B.np);
        A.take(&B.env_rstart, env_rstart.size()); A.take(&B.env_rows, env_rows.size()); A.take(&B.pose_edges_l, B.ne);
        in_bytes = A.off;
        A.take(&B.row_jobs, B.nf); A.take(&B.row_blks, B.nf); A.take(&B.job_off, B.nf + 1); A.take(&B.blk_off, B.nf + 1); A.take(&B.jtot, 2);
        A.take(&B.T, B.np); A.take(&B.Tb, B.np); A.take(&B.Xb, 3 * (size_t)B.nl); A.take(&B.err, 3 * (size_t)B.ne);
        A.take(&B.Hll, 9 * (size_t)B.nl); A.take(&B.bl, 3 * (size_t)B.nl); A.take(&B.Hpl, 18 * (size_t)B.ne); A.take(&B.W, 18 * (size_t)B.ne); A.take(&B.Hpp, 36 * (size_t)B.nf);
        A.take(&B.bp, B.NP); A.take(&B.S, (size_t)B.NP * B.NP); A.take(&B.coef, B.NP); A.take(&B.xp, B.NP); A.take(&B.xsol, B.NP); A.take(&B.xl, 3 * (size_t)B.nl);
        A.take(&B.Dinv, 9 * (size_t)B.nl); A.take(&B.dwork, B.NP); A.take(&B.partial, B.nblk_v);
        { double *blk = nullptr; A.take(&blk, 1 + (size_t)B.nblk_v + B.nblk_e); B.ok = (int *)blk; B.part_scale = blk ? blk + 1 : nullptr; B.part_chi = blk ? blk + 1 + B.nblk_v : nullptr; }
        A.take(&B.pt_active, B.nl); A.take(&derase, B.ne); A.take(&B.env_S2, env_nsep * env_nsep + env_nsep + 1);
        A.take(&B.jobs, jobs_cap); A.take(&B.blk_start, jobs_cap + 1); A.take(&B.Linv, (size_t)((B.NP + SGX_NB - 1) / SGX_NB) * SGX_NB * SGX_NB);
        const size_t total = A.off;
        A.base = save;
        if (pass == 0) { if ((rc = A.reserve(total)) != SGX_OK) return rc; }
    }
    {
        char *stage = stage_reserve(in_bytes); if (!stage) return SGX_ERR_NOMEM;       // (the 256-byte alignment gaps between the arrays are never read)
        char *base = g_arena.base;
        auto put = [&](const void *dst_dev, const void *src, size_t bytes) { memcpy(stage + ((const char *)dst_dev - base), src, bytes); };
        put(B.E, E.data(), sizeof(SgxBaEdge) * B.ne); put(B.X, Xd.data(), sizeof(double) * Xd.size());
        put(B.pt_start, pt_start.data(), 4 * (size_t)(B.nl + 1)); put(B.pt_edges, pt_edges.data(), 4 * (size_t)B.ne);
        put(B.pose_start, pose_start.data(), 4 * (size_t)(B.np + 1)); put(B.pose_edges, pose_edges.data(), 4 * (size_t)B.ne);
        put(B.pose_edges_l, pose_edges_l.data(), 4 * (size_t)B.ne);
        put(B.hidx, hidx.data(), 4 * (size_t)B.np); if (B.nf) put(B.free_pose, free_pose.data(), 4 * (size_t)B.nf);
        put(dTcw, P->poses, 64 * (size_t)B.np);
        if (!env_rstart.empty()) { put(B.env_rstart, env_rstart.data(), 4 * env_rstart.size()); put(B.env_rows, env_rows.data(), 4 * env_rows.size()); }
        if (mode == 0) put(dfixed, P->pose_fixed, B.np); else memset(stage + ((const char *)dfixed - base), 0, (size_t)B.np);           // mode 1: every keyframe is rewritten from its vertex (Optimizer.cc:200-214) -> flags stay 0
        SGX_CHECK_HIP(hipMemcpy(base, stage, in_bytes, hipMemcpyHostToDevice));
    }
    if (env_rstart.empty()) { B.env_rstart = nullptr; B.env_rows = nullptr; }
    B.env_nrows = env_rstart.empty() ? 0 : env_rstart.back();
    B.env_nA = env_nA; B.env_nB = env_nB; B.env_x2 = B.env_S2 + env_nsep * env_nsep;
    g_ba_last_plan[0] = B.env_rstart ? 1 : 0; g_ba_last_plan[1] = env_nB > 0 ? env_nA : (B.env_rstart ? (B.NP + SGX_NB - 1) / SGX_NB : 0); g_ba_last_plan[2] = env_nB; g_ba_last_plan[3] = (int)env_nsep;
    SGX_CHECK_HIP(hipMemsetAsync(B.xp, 0, sizeof(double) * (B.NP ? B.NP : 1), 0));
    SGX_CHECK_HIP(hipMemsetAsync(B.xsol, 0, sizeof(double) * (B.NP ? B.NP : 1), 0));
    SGX_CHECK_HIP(hipMemsetAsync(B.xl, 0, sizeof(double) * 3 * (size_t)B.nl, 0));
    SGX_CHECK_HIP(hipMemsetAsync(B.err, 0, sizeof(double) * 3 * (size_t)B.ne, 0));
    SGX_CHECK_HIP(hipMemsetAsync(B.Hpl, 0, sizeof(double) * 18 * (size_t)B.ne, 0));
    SGX_LAUNCH(k_ba_poses_in, dim3((B.np + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.np, dTcw, B.T);

    lap("arena + upload");
    int it1 = 0, it2 = 0; double chi1 = 0, chi2 = 0;
    std::vector<uint8_t> level1(B.ne, 0);
#ifdef SGX_EMU
    const bool jobs_on_host = true;
#else
    const bool jobs_on_host = g_ba_jobs_host != 0;
#endif
    rc = jobs_on_host ? build_jobs(B, pt_start, pt_edges, pose_start, pose_edges_l, E, nullptr, hidx, free_pose) : build_jobs_dev(B); if (rc != SGX_OK) return rc;
    lap("schur job list 1");
    rc = optimize(B, mode == 1 ? n_iterations : 5, &it1, &chi1); if (rc != SGX_OK) return rc;   // Optimizer.cc:659-660 / :187-188
    lap("optimize 1");
    if (mode == 0 && !stopped(B)) {                                                                      // :662-707
        SGX_LAUNCH(k_ba_classify, dim3(B.nblk_e), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.ne, B.E, B.T, B.X, B.err, 0, derase);
        if (!jobs_on_host) { rc = build_jobs_dev(B); if (rc != SGX_OK) return rc; }      // the active edge set changed: the kernels read the new levels in place
        else {   // mirror the new levels on the host to rebuild the Schur job list
            std::vector<SgxBaEdge> Eh(B.ne);
            SGX_CHECK_HIP(hipMemcpy(Eh.data(), B.E, sizeof(SgxBaEdge) * B.ne, hipMemcpyDeviceToHost));
            for (int k = 0; k < B.ne; k++) level1[k] = (Eh[k].flags & 2) ? 1 : 0;
            rc = build_jobs(B, pt_start, pt_edges, pose_start, pose_edges_l, E, &level1, hidx, free_pose); if (rc != SGX_OK) return rc;
        }
        lap("classify + job list 2");
        rc = optimize(B, 10, &it2, &chi2); if (rc != SGX_OK) return rc;
        lap("optimize 2");
    }
    if (mode == 0) SGX_LAUNCH(k_ba_classify, dim3(B.nblk_e), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.ne, B.E, B.T, B.X, B.err, 1, derase);   // :709-742
    SGX_LAUNCH(k_ba_poses_out, dim3((B.np + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.np, dfixed, B.T, dTcw);
    SGX_CHECK_HIP(hipGetLastError());
    if (mode == 0) SGX_CHECK_HIP(hipMemcpy(edge_erase, derase, B.ne, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(P->poses, dTcw, 64 * (size_t)B.np, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(Xd.data(), B.X, sizeof(double) * Xd.size(), hipMemcpyDeviceToHost));
    for (int l = 0; l < B.nl; l++)                                                           // Converter::toCvMat(Vector3d), Optimizer.cc:771-777 / :216-236
        if (mode == 0 || pt_start[l + 1] > pt_start[l])                                      // BundleAdjustment: points without edges were removed from the graph (vbNotIncludedMP)
            for (int c = 0; c < 3; c++) P->points[3 * (size_t)l + c] = (float)Xd[3 * (size_t)l + c];
    if (stats) { stats->iterations_first = it1; stats->iterations_second = it2; stats->chi2_first = chi1; stats->chi2_second = chi2; stats->free_poses = B.nf; }
    lap("classify + download");
    return SGX_OK;
}

extern "C" int sgx_local_bundle_adjustment(const sgx_ba_problem *P, const sgx_camera *cam, const volatile int32_t *stop_flag,
                                           uint8_t *edge_erase, sgx_ba_stats *stats)
{
    return ba_run(P, cam, stop_flag, edge_erase, stats, 0, 0, 1);
}

extern "C" int sgx_bundle_adjustment(const sgx_ba_problem *P, const sgx_camera *cam, int n_iterations, const volatile int32_t *stop_flag, int robust, sgx_ba_stats *stats)
{
    if (n_iterations < 0) return SGX_ERR_INVALID;
    return ba_run(P, cam, stop_flag, nullptr, stats, 1, n_iterations, robust);
}


// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// Optimizer::OptimizeEssentialGraph: the optimisation (Optimizer.cc:794, :961-962) on a flattened pose graph; kernels in sgx_eg_kernels.h
// ---------------------------------------------------------------------------------------------------------------------------------------------------------
namespace {
struct DevBufs {                                       // plain hipMalloc'ed buffers of one call (loop closing is rare: no arena)
    std::vector<void *> p;
    ~DevBufs() { for (void *q : p) (void)hipFree(q); }
    template <class T> int get(T **out, size_t n) { void *q = nullptr; if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return SGX_ERR_NOMEM; p.push_back(q); *out = (T *)q; return SGX_OK; }
    template <class T> int put(T **out, const T *src, size_t n) { int rc = get(out, n); if (rc != SGX_OK) return rc; if (n && hipMemcpy(*out, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return SGX_ERR_DEVICE; return SGX_OK; }
};
}

extern "C" int sgx_optimize_essential_graph(int nv, const double *S_in, const uint8_t *fixed, int ne, const int32_t *e_i, const int32_t *e_j, const double *e_meas,
                                            int fix_scale, int iterations, double *S_out, double *stats)
{
    if (nv < 0 || ne < 0 || iterations < 0 || (nv > 0 && (!S_in || !fixed || !S_out)) || (ne > 0 && (!e_i || !e_j || !e_meas))) return SGX_ERR_INVALID;
    if (stats) { stats[0] = 0; stats[1] = 0; stats[2] = 0; }
    for (int k = 0; k < ne; k++) if (e_i[k] < 0 || e_i[k] >= nv || e_j[k] < 0 || e_j[k] >= nv || e_i[k] == e_j[k]) return SGX_ERR_INVALID;
    if (nv > 0 && S_out != S_in) memcpy(S_out, S_in, sizeof(double) * 8 * (size_t)nv);
    std::vector<int> hidx((size_t)(nv > 0 ? nv : 1), -1);
    int nf = 0;
    for (int v = 0; v < nv; v++) hidx[(size_t)v] = fixed[v] ? -1 : nf++;
    const int NP = 7 * nf;
    if (NP == 0 || ne == 0 || iterations == 0) return SGX_OK;
    if (NP > SGX_BA_MAX_DENSE) return SGX_ERR_UNSUPPORTED;
    // incidence lists of the free vertices and the groups of edges that share an unordered vertex pair, both in edge order
    std::vector<int> inc_start((size_t)nf + 1, 0), inc_edge; std::vector<uint8_t> inc_side;
    for (int k = 0; k < ne; k++) { if (hidx[(size_t)e_i[k]] >= 0) inc_start[(size_t)hidx[(size_t)e_i[k]] + 1]++; if (hidx[(size_t)e_j[k]] >= 0) inc_start[(size_t)hidx[(size_t)e_j[k]] + 1]++; }
    for (int h = 0; h < nf; h++) inc_start[(size_t)h + 1] += inc_start[(size_t)h];
    inc_edge.resize((size_t)inc_start[(size_t)nf] > 0 ? (size_t)inc_start[(size_t)nf] : 1); inc_side.resize(inc_edge.size());
    { std::vector<int> fill((size_t)nf, 0);
      for (int k = 0; k < ne; k++) for (int side = 0; side < 2; side++) { const int h = hidx[(size_t)(side ? e_j[k] : e_i[k])]; if (h < 0) continue;
          const size_t q = (size_t)inc_start[(size_t)h] + (size_t)fill[(size_t)h]++; inc_edge[q] = k; inc_side[q] = (uint8_t)side; } }
    std::vector<int> order; order.reserve((size_t)ne);
    for (int k = 0; k < ne; k++) if (hidx[(size_t)e_i[k]] >= 0 && hidx[(size_t)e_j[k]] >= 0) order.push_back(k);
    auto lo_of = [&](int k) { return std::min(hidx[(size_t)e_i[k]], hidx[(size_t)e_j[k]]); };
    auto hi_of = [&](int k) { return std::max(hidx[(size_t)e_i[k]], hidx[(size_t)e_j[k]]); };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return lo_of(x) != lo_of(y) ? lo_of(x) < lo_of(y) : hi_of(x) < hi_of(y); });
    std::vector<int> pair_start, pair_lo, pair_hi, pair_edge; std::vector<uint8_t> pair_flip;
    for (size_t q = 0; q < order.size(); q++) {
        const int k = order[q];
        if (q == 0 || lo_of(k) != lo_of(order[q - 1]) || hi_of(k) != hi_of(order[q - 1])) { pair_start.push_back((int)q); pair_lo.push_back(lo_of(k)); pair_hi.push_back(hi_of(k)); }
        pair_edge.push_back(k); pair_flip.push_back((uint8_t)(hidx[(size_t)e_i[k]] > hidx[(size_t)e_j[k]]));
    }
    pair_start.push_back((int)order.size());
    const int npair = (int)pair_lo.size();
    if (pair_edge.empty()) { pair_edge.push_back(0); pair_flip.push_back(0); }
    if (pair_lo.empty()) { pair_lo.push_back(0); pair_hi.push_back(0); }

    DevBufs D; int rc;
    double *dV, *dVb, *dM, *derr, *dblk, *dH, *dS, *db, *dbp, *dcoef, *dxp, *dxsol, *dLinv, *dpart; int *dei, *dej, *dhidx, *dok, *dis, *die, *dps, *dpl, *dph, *dpe; uint8_t *dside, *dflip;
    const int nblk_e = (ne + SGX_EG_THREADS - 1) / SGX_EG_THREADS, nblk_v = (nv + SGX_EG_THREADS - 1) / SGX_EG_THREADS, npart = std::max(nblk_e, nblk_v);
#define TRY(x) if ((rc = (x)) != SGX_OK) return rc
    TRY(D.put(&dV, S_in, 8 * (size_t)nv)); TRY(D.get(&dVb, 8 * (size_t)nv)); TRY(D.put(&dM, e_meas, 8 * (size_t)ne)); TRY(D.put(&dei, e_i, (size_t)ne)); TRY(D.put(&dej, e_j, (size_t)ne));
    TRY(D.put(&dhidx, hidx.data(), (size_t)nv)); TRY(D.get(&derr, 7 * (size_t)ne)); TRY(D.get(&dblk, (size_t)SGX_EG_BLK * ne));
    TRY(D.get(&dH, (size_t)NP * NP)); TRY(D.get(&dS, (size_t)NP * NP)); TRY(D.get(&db, (size_t)NP)); TRY(D.get(&dbp, (size_t)NP)); TRY(D.get(&dcoef, (size_t)NP)); TRY(D.get(&dxp, (size_t)NP)); TRY(D.get(&dxsol, (size_t)NP));
    TRY(D.get(&dLinv, (size_t)((NP + SGX_NB - 1) / SGX_NB) * SGX_NB * SGX_NB)); TRY(D.get(&dok, 4)); TRY(D.get(&dpart, (size_t)npart));
    TRY(D.put(&dis, inc_start.data(), inc_start.size())); TRY(D.put(&die, inc_edge.data(), inc_edge.size())); TRY(D.put(&dside, inc_side.data(), inc_side.size()));
    TRY(D.put(&dps, pair_start.data(), pair_start.size())); TRY(D.put(&dpl, pair_lo.data(), pair_lo.size())); TRY(D.put(&dph, pair_hi.data(), pair_hi.size()));
    TRY(D.put(&dpe, pair_edge.data(), pair_edge.size())); TRY(D.put(&dflip, pair_flip.data(), pair_flip.size()));
#undef TRY
    SGX_CHECK_HIP(hipMemset(dxp, 0, sizeof(double) * (size_t)NP)); SGX_CHECK_HIP(hipMemset(dxsol, 0, sizeof(double) * (size_t)NP));
    std::vector<double> hp((size_t)npart);
    auto chi2_at = [&](const double *V, double *chi) -> int {
        SGX_LAUNCH(k_eg_errors, dim3(nblk_e), dim3(SGX_EG_THREADS), (sgx_stream_t)0, ne, dei, dej, dM, V, derr, dpart);
        SGX_CHECK_HIP(hipMemcpy(hp.data(), dpart, sizeof(double) * (size_t)nblk_e, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < nblk_e; i++) s += hp[(size_t)i]; *chi = s; return SGX_OK;
    };
    double lambda = -1, ni = 2; int nBadLM = 0, iters = 0; double chi_first = 0, chi_last = 0;
    for (int it = 0; it < iterations; it++) {
        double currentChi = 0; if ((rc = chi2_at(dV, &currentChi)) != SGX_OK) return rc;
        if (it == 0) chi_first = currentChi;
        double tempChi = currentChi; const double iniChi = currentChi;
        SGX_LAUNCH(k_eg_linearize, dim3(nblk_e), dim3(SGX_EG_THREADS), (sgx_stream_t)0, ne, dei, dej, dM, dV, dhidx, fix_scale, derr, dblk);
        SGX_CHECK_HIP(hipMemsetAsync(dH, 0, sizeof(double) * (size_t)NP * NP, 0));
        SGX_LAUNCH(k_eg_assemble_diag, dim3(nf), dim3(64), (sgx_stream_t)0, NP, dis, die, dside, dblk, dH, db);
        if (npair > 0) SGX_LAUNCH(k_eg_assemble_pairs, dim3(npair), dim3(64), (sgx_stream_t)0, NP, dps, dpl, dph, dpe, dflip, dblk, dH);
        if (it == 0) { lambda = 1e-16; ni = 2; nBadLM = 0; }                  // computeLambdaInit with setUserLambdaInit(1e-16)
        double rho = 0; int qmax = 0;
        do {
            SGX_CHECK_HIP(hipMemcpyAsync(dVb, dV, sizeof(double) * 8 * (size_t)nv, hipMemcpyDeviceToDevice, 0));       // push
            { const int one = 1; SGX_CHECK_HIP(hipMemcpyAsync(dok, &one, 4, hipMemcpyHostToDevice, 0)); }
            const size_t n2 = (size_t)NP * NP; const int g = (int)std::min<size_t>((n2 + SGX_EG_THREADS - 1) / SGX_EG_THREADS, 8192);
            SGX_LAUNCH(k_eg_damp, dim3(g), dim3(SGX_EG_THREADS), (sgx_stream_t)0, NP, dH, db, lambda, dS, dbp, dcoef);
            const double *xsol = dxp;
            { const Chol C = { NP, dS, dLinv, dbp, dcoef, dxp, dxsol, dok }; if ((rc = chol_factor_solve(C, &xsol)) != SGX_OK) return rc; }
            SGX_LAUNCH(k_eg_update, dim3(nblk_v), dim3(SGX_EG_THREADS), (sgx_stream_t)0, nv, dhidx, xsol, db, lambda, fix_scale, dok, dVb, dV, dpart);
            SGX_CHECK_HIP(hipMemcpy(hp.data(), dpart, sizeof(double) * (size_t)nblk_v, hipMemcpyDeviceToHost));
            double scale = 0; for (int i = 0; i < nblk_v; i++) scale += hp[(size_t)i];
            int ok2 = 1; SGX_CHECK_HIP(hipMemcpy(&ok2, dok, 4, hipMemcpyDeviceToHost));
            if ((rc = chi2_at(dV, &tempChi)) != SGX_OK) return rc;
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            scale += 1e-3; rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                const double r21 = 2 * rho - 1;
                double alpha = 1. - r21 * r21 * r21; alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                lambda *= (alpha > 1. / 3. ? alpha : 1. / 3.); ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                SGX_CHECK_HIP(hipMemcpyAsync(dV, dVb, sizeof(double) * 8 * (size_t)nv, hipMemcpyDeviceToDevice, 0));   // pop
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        iters = it + 1; chi_last = currentChi;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) break;
    }
    SGX_CHECK_HIP(hipMemcpy(S_out, dV, sizeof(double) * 8 * (size_t)nv, hipMemcpyDeviceToHost));
    if (stats) { stats[0] = iters; stats[1] = chi_first; stats[2] = chi_last; }
    return SGX_OK;
}

extern "C" int sgx_correct_map_points(int n, const float *xw, const int32_t *ref, int nv, const double *Srw, const double *corrected_Swr, float *xw_out)
{
    if (n < 0 || nv < 0 || (n > 0 && (!xw || !ref || !Srw || !corrected_Swr || !xw_out))) return SGX_ERR_INVALID;
    if (n == 0) return SGX_OK;
    for (int i = 0; i < n; i++) if (ref[i] < 0 || ref[i] >= nv) return SGX_ERR_INVALID;
    DevBufs D; int rc; float *dx, *dout; int *dref; double *da, *dc;
    if ((rc = D.put(&dx, xw, 3 * (size_t)n)) != SGX_OK || (rc = D.put(&dref, ref, (size_t)n)) != SGX_OK || (rc = D.put(&da, Srw, 8 * (size_t)nv)) != SGX_OK ||
        (rc = D.put(&dc, corrected_Swr, 8 * (size_t)nv)) != SGX_OK || (rc = D.get(&dout, 3 * (size_t)n)) != SGX_OK) return rc;
    SGX_LAUNCH(k_eg_correct_points, dim3((n + SGX_EG_THREADS - 1) / SGX_EG_THREADS), dim3(SGX_EG_THREADS), (sgx_stream_t)0, n, dx, dref, da, dc, dout);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(xw_out, dout, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    return SGX_OK;
}
