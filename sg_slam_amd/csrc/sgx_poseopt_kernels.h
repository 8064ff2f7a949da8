// sgx_poseopt_kernels.h — HIP kernel for Optimizer::PoseOptimization (fp64 Levenberg-Marquardt
// over one SE3 pose and N unary reprojection edges), one 64-lane wave per frame.  Phase style.
// Reference behaviour: src/sg-slam/src/Optimizer.cc:239-451 and the vendored g2o it drives
// (G = src/sg-slam/Thirdparty/g2o/g2o; cited inline; the CPU restatement is oracle/poseopt_oracle.c).
#pragma once
#include "sgx_rt.h"
#include "sgx_types.h"
#include "sgx_block.h"

#define SGX_PO_CAP 1280
#define SGX_PO_THREADS 256
#define SGX_PO_NRED 27            /* 21 upper-triangular H entries + 6 b entries (odd row stride: conflict-free LDS writes) */
#define SGX_PO_CG 16              /* groups of the chi2 reduction */

#include "sgx_se3.h"

// Sum of a double over groups of 8 consecutive lanes, as the pairwise tree ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); every lane of the group receives it.  On the device
// three DPP steps (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror) on the two halves of the value; the emulator forms the same tree from the threads' stored values.
#ifndef SGX_EMU
template <int CTRL> SGX_DEV double sgx_dpp_add_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return v + __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, true));
}
SGX_DEV double sgx_sum8_f64(double v) { v = sgx_dpp_add_f64<0xB1>(v); v = sgx_dpp_add_f64<0x4E>(v); return sgx_dpp_add_f64<0x141>(v); }
#endif
SGX_DEV double sgx_tree8_f64(const double *a, int stride) { return ((a[0] + a[stride]) + (a[2 * stride] + a[3 * stride])) + ((a[4 * stride] + a[5 * stride]) + (a[6 * stride] + a[7 * stride])); }

// ---------------------------------------------------------------------------------------------
// k_pose_opt: one 256-thread workgroup per frame.  Edge e <-> keypoint i with a map point (ascending i, as the
// reference inserts them, Optimizer.cc:280-360).  Threads own edges e = tid, tid+256, ...; the
// 6x6 system, the chi2 sums and the LM control flow are wave-uniform (each lane evaluates the same
// scalar code on values reduced through LDS), so the kernel follows the reference's accept/reject,
// lambda schedule and stop rules statement for statement (levenberg.cpp:61-164).
// Sums over edges are reduced in a fixed order (thread-strided partials, interleaved groups of rows, then the groups):
// deterministic, and within ~1e-15 relative of the reference's sequential order.
// Measured with clock64 on MI355X (tools/ubench/f64_issue.hip): a wave issues a dependent fp64 FMA every 7.3 cycles and, with eight
// independent chains, still only every 5.2; unrolling the edge loops four-fold bought 15 % per edge but paid it back in idle tail slots, and a
// 512-thread variant (two waves per SIMD) was slower at every phase.  So the kernel's time is its fp64 instruction count: the per-edge code is
// branch-free with a single division (mono / stereo, excluded edges and the Huber kernel are selects), multiply-adds fuse (sgx_poseopt.cpp)
// and the wave-uniform solver multiplies by reciprocals.  Phase split of one launch (800 points, 54 LM trials): residuals 45 %, solver +
// exp 25 %, linearisation 21 %, classification and set-up 9 %.
// mp_index (optional): map point of keypoint i is table row mp_index[i] (-1 = none); otherwise has_mp/xw are per keypoint.
// ---------------------------------------------------------------------------------------------
// NTT = threads per frame: 256 (4 waves: lowest latency per frame) or 64 (one wave per frame: no inter-wave barriers to wait on and 4x fewer
// waves for the redundant serial part, so large batches of frames pack four to a CU and finish sooner in aggregate).  Same arithmetic order
// within a thread; the partial sums are regrouped (NTT/32 groups of 32), i.e. results differ only in the last bits of the reductions.
template <int NTT>
SGX_KERNEL(NTT) k_pose_opt(int cap, const uint8_t *keys_raw, const float *uright, const int *n_kp,
                                      const int *mp_index, const uint8_t *has_mp, const float *xw, int xw_pitch,
                                      SgxScales inv_sigma2, SgxCam cam, float *Tcw, uint8_t *outlier, int *n_inliers)
{
    SGX_LDS float e_obs[SGX_PO_CAP * 3], e_xw[SGX_PO_CAP * 3], e_info[SGX_PO_CAP];
    SGX_LDS double e_err[SGX_PO_CAP * 3];
    SGX_LDS uint16_t e_kp[SGX_PO_CAP];
    SGX_LDS uint8_t e_flags[SGX_PO_CAP];          // bit0 stereo, bit1 level==1 (excluded), bit3 outlier flag
    // Reduction of the per-thread sums (round 4): first over groups of 8 lanes in registers (sgx_sum8_f64), then R8 = NTT / 8 rows through LDS in NG groups, then the groups.
    // `part` used to hold one row per THREAD (55 KB at 256 threads): with it the kernel took 128 KB of LDS, i.e. ONE workgroup = one wave per SIMD per CU — and a wave issues
    // a dependent fp64 FMA only every 7.3 cycles.  At 80 KB two frames share a CU and fill each other's issue gaps.
    constexpr int R8 = NTT / 8, NG = R8 >= 8 ? R8 / 8 : 1, RPG = R8 / NG;      // rows, groups, rows per group (rows j, j + NG, ... belong to group j)
    SGX_LDS double part[R8 * SGX_PO_NRED];
    SGX_LDS double part2[SGX_PO_NRED * NG];
    SGX_LDS double red[SGX_PO_NRED];
    SGX_LDS double chi_part[R8], chi_grp[SGX_PO_CG];
#ifdef SGX_EMU
    static thread_local double part_thr[NTT * SGX_PO_NRED], chi_thr[NTT];         // the emulator's threads run one after the other: their values wait here for the 8-lane tree
#endif
    SGX_LDS int scan[NTT];
    SGX_LDS int s_ne, s_nbad;

    // one workgroup per frame on the latency-critical tracking stream: out-prioritise the throughput kernels (extraction of the next frame)
    // that share the CU, whose thousands of waves do not care about a few lost issue slots
    SGX_WAVE_PRIORITY(3);
    const int f = (int)blockIdx.x;
    const int N = min(n_kp[f], cap);
    const int NT = NTT;
    const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);   // Optimizer.cc:272-273 (float)
    const double fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy, bf = cam.bf;

    // ---- build the edge list in ascending keypoint order: per-thread contiguous chunks + block scan
    const int CH = (N + NT - 1) / NT;
    SGX_THREADS_BEGIN(tid)
    int c = 0;
    for (int i = tid * CH; i < min(N, (tid + 1) * CH); i++) {
        const size_t o = (size_t)f * cap + i;
        c += mp_index ? (mp_index[o] >= 0) : (has_mp[o] != 0);
    }
    scan[tid] = c;
    for (int i = tid; i < cap; i += NT) outlier[(size_t)f * cap + i] = 0;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    sgx_block_exclusive_scan_i32(scan, NT, &s_ne, tid);
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    int ne_ = scan[tid];
    for (int i = tid * CH; i < min(N, (tid + 1) * CH); i++) {
        const size_t o = (size_t)f * cap + i;
        int src = -1;
        if (mp_index) { const int m = mp_index[o]; if (m >= 0) src = m; }
        else if (has_mp[o]) src = i;
        if (src < 0) continue;
        const float *kp = (const float *)(keys_raw + o * 28);
        const float ur = uright[o];
        e_obs[3 * ne_] = kp[0]; e_obs[3 * ne_ + 1] = kp[1]; e_obs[3 * ne_ + 2] = ur;
        const float *X = xw + ((size_t)f * xw_pitch + src) * 3;
        e_xw[3 * ne_] = X[0]; e_xw[3 * ne_ + 1] = X[1]; e_xw[3 * ne_ + 2] = X[2];
        e_info[ne_] = inv_sigma2.s[((const int *)kp)[5]];
        e_kp[ne_] = (uint16_t)i;
        e_flags[ne_] = (uint8_t)(ur < 0 ? 0 : 1);              // mono iff mvuRight<0 (Optimizer.cc:286)
        ne_++;
    }
    SGX_THREADS_END
    SGX_SYNC();
    const int ne = s_ne;
    if (ne < 3) {                                                   // Optimizer.cc:364-365
        SGX_THREADS_BEGIN(tid) if (tid == 0) n_inliers[f] = 0; SGX_THREADS_END
        return;
    }

    SgxSE3 est;
    float T0[16];
    for (int i = 0; i < 16; i++) T0[i] = Tcw[16 * f + i];
    int nBad = 0;

#ifndef SGX_EMU
#define SGX_PO_STORE_CHI8(chi_) { const double c8_ = sgx_sum8_f64(chi_); if ((tid & 7) == 0) chi_part[tid >> 3] = c8_; }
#define SGX_PO_EMU_CHI8()
#else
#define SGX_PO_STORE_CHI8(chi_) chi_thr[tid] = (chi_);
#define SGX_PO_EMU_CHI8() for (int r_ = 0; r_ < R8; r_++) chi_part[r_] = sgx_tree8_f64(chi_thr + 8 * r_, 1);
#endif
// evaluates the errors of the edges at `est` and the robust chi2 of the active (level-0) ones -> chiTot.  Excluded edges are evaluated too
// (their e_err is recomputed by the classification below before it is next read) but add an exact zero.
#define SGX_PO_ERRORS()                                                                                     \
    SGX_THREADS_BEGIN(tid)                                                                                  \
    double chi = 0;                                                                                         \
    for (int e = tid; e < ne; e += NT) {                                                                    \
        const int fl = e_flags[e];                                                                          \
        double er[3];                                                                                       \
        sgx_po_residual(est, e_xw + 3 * e, e_obs + 3 * e, (fl & 1) != 0, fx, fy, cx, cy, bf, er);            \
        e_err[3 * e] = er[0]; e_err[3 * e + 1] = er[1]; e_err[3 * e + 2] = er[2];                           \
        const double c2 = sgx_po_chi2(er, (double)e_info[e], 1);                                            \
        double r0 = c2;                                                                                     \
        if (robust) r0 = sgx_huber_rho0((fl & 2) ? 0.0 : c2, (fl & 1) ? deltaStereo : deltaMono);           \
        chi += (fl & 2) ? 0.0 : r0;                                                                         \
    }                                                                                                       \
    SGX_PO_STORE_CHI8(chi)                                                                                  \
    SGX_THREADS_END                                                                                         \
    SGX_PO_EMU_CHI8()                                                                                       \
    SGX_SYNC();                                                                                             \
    SGX_THREADS_BEGIN(tid)                                                                                  \
    if (tid < SGX_PO_CG) { double s = 0; for (int l = 0; l * SGX_PO_CG + tid < R8; l++) s += chi_part[l * SGX_PO_CG + tid]; chi_grp[tid] = s; } \
    SGX_THREADS_END                                                                                         \
    SGX_SYNC();                                                                                             \
    { double s = 0; for (int l = 0; l < SGX_PO_CG; l++) s += chi_grp[l]; chiTot = s; }

    for (int round = 0; round < 4; round++) {
        sgx_se3_from_cv(T0, est);                                   // Optimizer.cc:377: every round restarts from pFrame->mTcw
        double lambda = -1, ni = 2; int nBadLM = 0;
        const bool robust = round < 3;                  // every edge carries the Huber kernel until the third classification drops it (Optimizer.cc:436)
        double chiTot = 0;
        bool fresh = false; double freshChi = 0;        // e_err / chi already evaluated at `est` by an accepted trial
        for (int it = 0; it < 10; it++) {
            if (!fresh) { SGX_PO_ERRORS() freshChi = chiTot; }    // computeActiveErrors + activeRobustChi2 (levenberg.cpp:73-80)
            double currentChi = freshChi;
            double tempChi = currentChi;
            const double iniChi = currentChi;
            // ---- buildSystem: b -= rho1 * J^T (Omega e), H += J^T (rho1 Omega) J   (base_unary_edge.hpp:43-72)
            SGX_THREADS_BEGIN(tid)
            double acc[27];
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] = 0;
            for (int e = tid; e < ne; e += NT) {
                const int fl = e_flags[e];
                const bool stereo = (fl & 1) != 0, active = !(fl & 2);
                const double Xd[3] = { (double)e_xw[3 * e], (double)e_xw[3 * e + 1], (double)e_xw[3 * e + 2] };
                double p[3]; sgx_se3_map(est, Xd, p);
                // an excluded edge contributes exact zeros: zero weight and residual on a harmless point, so that no inf / NaN can leak in
                const double x = active ? p[0] : 0.0, y = active ? p[1] : 0.0, invz = 1.0 / (active ? p[2] : 1.0), invz_2 = invz * invz;
                double J[3][6];                                      // types_six_dof_expmap.cpp:266-288, 335-364
                J[0][0] = x * y * invz_2 * fx; J[0][1] = -(1 + (x * x * invz_2)) * fx; J[0][2] = y * invz * fx;
                J[0][3] = -invz * fx; J[0][4] = 0; J[0][5] = x * invz_2 * fx;
                J[1][0] = (1 + y * y * invz_2) * fy; J[1][1] = -x * y * invz_2 * fy; J[1][2] = -x * invz * fy;
                J[1][3] = 0; J[1][4] = -invz * fy; J[1][5] = y * invz_2 * fy;
                // mono edge: the third row does not exist (adds exact zeros below)
                J[2][0] = stereo ? J[0][0] - bf * y * invz_2 : 0.0; J[2][1] = stereo ? J[0][1] + bf * x * invz_2 : 0.0; J[2][2] = stereo ? J[0][2] : 0.0;
                J[2][3] = stereo ? J[0][3] : 0.0; J[2][4] = 0; J[2][5] = stereo ? J[0][5] - bf * invz_2 : 0.0;
                const double info = active ? (double)e_info[e] : 0.0;
                const double er[3] = { active ? e_err[3 * e] : 0.0, active ? e_err[3 * e + 1] : 0.0, (active && stereo) ? e_err[3 * e + 2] : 0.0 };
                double rho1 = 1.0;
                if (robust) rho1 = sgx_huber_rho1(sgx_po_chi2(er, info, 1), stereo ? deltaStereo : deltaMono);
                const double w = rho1 * info;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    const double s = J[0][a] * (info * er[0]) + J[1][a] * (info * er[1]) + J[2][a] * (info * er[2]);
                    acc[21 + a] -= rho1 * s;
#pragma unroll
                    for (int c = 0; c < 6; c++) if (c >= a) {
                        const double h = J[0][a] * w * J[0][c] + J[1][a] * w * J[1][c] + J[2][a] * w * J[2][c];
                        acc[a * 6 - (a * (a - 1)) / 2 + (c - a)] += h;
                    }
                }
            }
#ifndef SGX_EMU
#pragma unroll
            for (int k = 0; k < 27; k++) { const double s8 = sgx_sum8_f64(acc[k]); if ((tid & 7) == 0) part[(tid >> 3) * SGX_PO_NRED + k] = s8; }
#else
            for (int k = 0; k < 27; k++) part_thr[tid * SGX_PO_NRED + k] = acc[k];
#endif
            SGX_THREADS_END
#ifdef SGX_EMU
            for (int r_ = 0; r_ < R8; r_++) for (int k = 0; k < 27; k++) part[r_ * SGX_PO_NRED + k] = sgx_tree8_f64(part_thr + (8 * r_) * SGX_PO_NRED + k, SGX_PO_NRED);
#endif
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            if (tid < 27 * NG) { const int c = tid / NG, j = tid - c * NG; double s = 0; for (int l = 0; l < RPG; l++) s += part[(l * NG + j) * SGX_PO_NRED + c]; part2[c * NG + j] = s; }
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            if (tid < 27) { double s = 0; for (int l = 0; l < NG; l++) s += part2[tid * NG + l]; red[tid] = s; }
            SGX_THREADS_END
            SGX_SYNC();
            double H[6][6], b[6];
#pragma unroll
            for (int a = 0; a < 6; a++) {
#pragma unroll
                for (int c = 0; c < 6; c++) if (c >= a) { const double v = sgx_uniform_f64(red[a * 6 - (a * (a - 1)) / 2 + (c - a)]); H[a][c] = v; H[c][a] = v; }
                b[a] = sgx_uniform_f64(red[21 + a]);
            }
            if (it == 0) {                                           // computeLambdaInit, levenberg.cpp:166-180 (tau = 1e-5)
                double maxd = 0;
#pragma unroll
                for (int j = 0; j < 6; j++) if (fabs(H[j][j]) > maxd) maxd = fabs(H[j][j]);
                lambda = 1e-5 * maxd; ni = 2; nBadLM = 0;
            }
            double rho = 0; int qmax = 0;
            do {
                const SgxSE3 backup = est;                           // push
                double Hl[6][6];
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int c = 0; c < 6; c++) Hl[a][c] = H[a][c] + (a == c ? lambda : 0.0);
                }
                double x[6] = { 0, 0, 0, 0, 0, 0 };
                const bool ok2 = sgx_ldlt6_solve(Hl, b, x);
                SgxSE3 ex; sgx_se3_exp(x, ex);
                SgxSE3 upd; sgx_se3_mul(ex, est, upd); est = upd;     // VertexSE3Expmap::oplusImpl
                SGX_PO_ERRORS()
                tempChi = chiTot;
                if (!ok2) tempChi = 1.7976931348623157e308;
                rho = currentChi - tempChi;
                double scale = 0;
#pragma unroll
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    const double r21 = 2 * rho - 1;
                    double alpha = 1. - r21 * r21 * r21;
                    alpha = alpha < 2. / 3. ? alpha : 2. / 3.;
                    const double sf = alpha > 1. / 3. ? alpha : 1. / 3.;
                    lambda *= sf; ni = 2; currentChi = tempChi; fresh = true; freshChi = tempChi;
                } else { lambda *= ni; ni *= 2; est = backup; fresh = false; }   // pop: the edges keep the rejected trial's errors
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) break;                       // Terminate
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        // ---- classify edges (Optimizer.cc:383-438)
        SGX_THREADS_BEGIN(tid)
        if (tid == 0) s_nbad = 0;
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        int bad = 0;
        for (int e = tid; e < ne; e += NT) {
            int fl = e_flags[e];
            const int stereo = fl & 1;
            if (fl & 8) sgx_po_edge_error(est, e_xw + 3 * e, e_obs + 3 * e, stereo, fx, fy, cx, cy, bf, e_err + 3 * e);
            const float chi2 = (float)sgx_po_chi2(e_err + 3 * e, (double)e_info[e], stereo);
            if (chi2 > (stereo ? 7.815f : 5.991f)) { fl |= (8 | 2); bad++; } else { fl &= ~(8 | 2); }
            e_flags[e] = (uint8_t)fl;
        }
        if (bad) sgx_atomic_add(&s_nbad, bad);
        SGX_THREADS_END
        SGX_SYNC();
        nBad = s_nbad;
        if (ne < 10) break;                                          // optimizer.edges().size()<10
    }
#undef SGX_PO_ERRORS
    SGX_THREADS_BEGIN(tid)
    for (int e = tid; e < ne; e += NT) outlier[(size_t)f * cap + e_kp[e]] = (e_flags[e] & 8) ? 1 : 0;
    if (tid == 0) { float To[16]; sgx_se3_to_cv(est, To); for (int i = 0; i < 16; i++) Tcw[16 * f + i] = To[i]; n_inliers[f] = ne - nBad; }
    SGX_THREADS_END
}
