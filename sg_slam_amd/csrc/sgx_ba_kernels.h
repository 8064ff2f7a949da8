// sgx_ba_kernels.h — HIP kernels for Optimizer::LocalBundleAdjustment (fp64): edge errors, J^T J / J^T r block
// accumulation, Schur complement on the 3x3 landmark blocks, dense LDL^T of the reduced camera system,
// landmark back-substitution, manifold update.  Phase style (sgx_rt.h).
// Reference behaviour: src/sg-slam/src/Optimizer.cc:453-778 and vendored g2o (G = src/sg-slam/Thirdparty/g2o/g2o):
// block_solver.hpp:354-604, base_binary_edge.hpp:55-120, types_six_dof_expmap.{h,cpp}; CPU restatement: oracle/localba_oracle.c.
#pragma once
#include "sgx_rt.h"
#include "sgx_types.h"
#include "sgx_se3.h"                 // SgxSE3 helpers (SE3Quat restatement), sgx_huber, sgx_po_chi2

#define SGX_BA_THREADS 256
#define SGX_BA_MAX_DENSE 24576       /* largest reduced camera system (6 * free poses) factorised densely (4.8 GB of fp64) */

// edge record: pose index, point index, flags (bit0 stereo, bit1 level==1, bit2 Huber on)
struct SgxBaEdge { int pose, point, flags; float obs[3]; float info; };

// EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ::computeError (types_six_dof_expmap.h:90-94,122-126; .cpp:141-157): stereo invz is a float
SGX_DEV void sgx_ba_edge_error(const SgxSE3 &T, const double *X, const SgxBaEdge &e, double fx, double fy, double cx, double cy, double bf, double err[3])
{
    double p[3]; sgx_se3_map(T, X, p);
    if (!(e.flags & 1)) { err[0] = (double)e.obs[0] - (p[0] / p[2] * fx + cx); err[1] = (double)e.obs[1] - (p[1] / p[2] * fy + cy); err[2] = 0; }
    else {
        const float invz = (float)(1.0 / p[2]);
        const double r0 = p[0] * invz * fx + cx, r1 = p[1] * invz * fy + cy, r2 = r0 - bf * invz;
        err[0] = (double)e.obs[0] - r0; err[1] = (double)e.obs[1] - r1; err[2] = (double)e.obs[2] - r2;
    }
}

// k_ba_errors: computeActiveErrors + activeRobustChi2 (sparse_optimizer.cpp:61-114).  Per-block partial sums in
// partial[blockIdx.x] (summed in block order by the host: deterministic).
SGX_KERNEL(SGX_BA_THREADS) k_ba_errors(int ne, const SgxBaEdge *E, const SgxSE3 *T, const double *X, SgxCam cam, double dMono, double dStereo,
                                       double *err, double *partial)
{
    SGX_LDS double red[SGX_BA_THREADS];
    SGX_THREADS_BEGIN(tid)
    const int k = (int)blockIdx.x * SGX_BA_THREADS + tid;
    double chi = 0;
    if (k < ne) {
        const SgxBaEdge e = E[k];
        if (!(e.flags & 2)) {
            double er[3];
            sgx_ba_edge_error(T[e.pose], X + 3 * (size_t)e.point, e, cam.fx, cam.fy, cam.cx, cam.cy, cam.bf, er);
            err[3 * (size_t)k] = er[0]; err[3 * (size_t)k + 1] = er[1]; err[3 * (size_t)k + 2] = er[2];
            const double c2 = sgx_po_chi2(er, (double)e.info, e.flags & 1);
            if (e.flags & 4) { double r0, r1; sgx_huber(c2, (e.flags & 1) ? dStereo : dMono, &r0, &r1); chi = r0; } else chi = c2;
        }
    }
    red[tid] = chi;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { double s = 0; for (int i = 0; i < SGX_BA_THREADS; i++) s += red[i]; partial[blockIdx.x] = s; }
    SGX_THREADS_END
}

// Jacobians of one edge at the current estimates (EdgeSE3ProjectXYZ::linearizeOplus .cpp:103-139, stereo :188-234):
// A = d e / d point (3x3, row 2 zero for mono), Bj = d e / d pose (3x6)
SGX_DEV void sgx_ba_jacobians(const SgxSE3 &T, const double *X, int stereo, double fx, double fy, double bf, double A[3][3], double Bj[3][6])
{
    double p[3]; sgx_se3_map(T, X, p);
    const double *q = T.q;
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    const double R[3][3] = { { 1 - (tyy + tzz), txy - twz, txz + twy }, { txy + twz, 1 - (txx + tzz), tyz - twx }, { txz - twy, tyz + twx, 1 - (txx + tyy) } };
    const double x = p[0], y = p[1], z = p[2], z_2 = z * z;
    if (stereo) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            A[0][c] = -fx * R[0][c] / z + fx * x * R[2][c] / z_2;
            A[1][c] = -fy * R[1][c] / z + fy * y * R[2][c] / z_2;
            A[2][c] = A[0][c] - bf * R[2][c] / z_2;
        }
    } else {
        const double t02 = -x / z * fx, t12 = -y / z * fy;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            A[0][c] = -1. / z * (fx * R[0][c] + 0 * R[1][c] + t02 * R[2][c]);
            A[1][c] = -1. / z * (0 * R[0][c] + fy * R[1][c] + t12 * R[2][c]);
            A[2][c] = 0;
        }
    }
    Bj[0][0] = x * y / z_2 * fx; Bj[0][1] = -(1 + (x * x / z_2)) * fx; Bj[0][2] = y / z * fx; Bj[0][3] = -1. / z * fx; Bj[0][4] = 0; Bj[0][5] = x / z_2 * fx;
    Bj[1][0] = (1 + y * y / z_2) * fy; Bj[1][1] = -x * y / z_2 * fy; Bj[1][2] = -x / z * fy; Bj[1][3] = 0; Bj[1][4] = -1. / z * fy; Bj[1][5] = y / z_2 * fy;
    if (stereo) { Bj[2][0] = Bj[0][0] - bf * y / z_2; Bj[2][1] = Bj[0][1] + bf * x / z_2; Bj[2][2] = Bj[0][2]; Bj[2][3] = Bj[0][3]; Bj[2][4] = 0; Bj[2][5] = Bj[0][5] - bf / z_2; }
    else {
#pragma unroll
        for (int c = 0; c < 6; c++) Bj[2][c] = 0;
    }
}

// k_ba_linearize_points: one thread per landmark walks the landmark's edges (CSR by point, insertion order) and builds
// Hll (3x3), bl (3) and, per edge with a free pose, the off-diagonal block Hpl_e = B^T W A (6x3)
// (BaseBinaryEdge::constructQuadraticForm, base_binary_edge.hpp:55-120).
SGX_KERNEL(SGX_BA_THREADS) k_ba_linearize_points(int nl, const int *pt_start, const int *pt_edges, const SgxBaEdge *E, const SgxSE3 *T, const double *X,
                                                 const int *hidx, const double *err, SgxCam cam, double dMono, double dStereo,
                                                 double *Hll, double *bl, double *Hpl, uint8_t *pt_active)
{
    SGX_THREADS_BEGIN(tid)
    const int l = (int)blockIdx.x * SGX_BA_THREADS + tid;
    if (l < nl) {
        double H[9], b[3]; int active = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) H[i] = 0;
        b[0] = b[1] = b[2] = 0;
        const double Xl[3] = { X[3 * (size_t)l], X[3 * (size_t)l + 1], X[3 * (size_t)l + 2] };
        for (int q = pt_start[l]; q < pt_start[l + 1]; q++) {
            const int k = pt_edges[q];
            const SgxBaEdge e = E[k];
            double *hpl = Hpl + (size_t)k * 18;
            if (e.flags & 2) continue;
            active = 1;
            const int stereo = e.flags & 1;
            double A[3][3], Bj[3][6];
            sgx_ba_jacobians(T[e.pose], Xl, stereo, cam.fx, cam.fy, cam.bf, A, Bj);
            const double info = (double)e.info;
            const double er[3] = { err[3 * (size_t)k], err[3 * (size_t)k + 1], stereo ? err[3 * (size_t)k + 2] : 0.0 };
            double rho1 = 1.0;
            if (e.flags & 4) { double r0; sgx_huber(sgx_po_chi2(er, info, stereo), stereo ? dStereo : dMono, &r0, &rho1); }
            const double w = rho1 * info;
            const double om[3] = { -(info * er[0]) * rho1, -(info * er[1]) * rho1, -(info * er[2]) * rho1 };
#pragma unroll
            for (int a = 0; a < 3; a++) {
                b[a] += A[0][a] * om[0] + A[1][a] * om[1] + A[2][a] * om[2];
#pragma unroll
                for (int c = 0; c < 3; c++) H[3 * a + c] += A[0][a] * w * A[0][c] + A[1][a] * w * A[1][c] + A[2][a] * w * A[2][c];
            }
            if (hidx[e.pose] >= 0) {
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                    for (int c = 0; c < 3; c++) hpl[3 * a + c] = Bj[0][a] * w * A[0][c] + Bj[1][a] * w * A[1][c] + Bj[2][a] * w * A[2][c];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 9; i++) Hll[(size_t)l * 9 + i] = H[i];
        bl[3 * (size_t)l] = b[0]; bl[3 * (size_t)l + 1] = b[1]; bl[3 * (size_t)l + 2] = b[2];
        pt_active[l] = (uint8_t)active;
    }
    SGX_THREADS_END
}

// k_ba_linearize_poses: one workgroup per FREE pose; its threads stride over the pose's edges (CSR by pose), stage the
// per-edge contributions B^T W B (21 upper entries) and B^T omega_r (6) in registers, then reduce the Jacobian tiles
// through LDS in a fixed order -> Hpp (6x6) and bp (6).  No atomics: deterministic.
SGX_KERNEL(SGX_BA_THREADS) k_ba_linearize_poses(int np, const int *free_pose, const int *pose_start, const int *pose_edges, const SgxBaEdge *E, const SgxSE3 *T,
                                                const double *X, const double *err, SgxCam cam, double dMono, double dStereo, double *Hpp, double *bp)
{
    SGX_LDS double part[SGX_BA_THREADS * 27];
    SGX_LDS double part2[27 * 8];
    const int hp = (int)blockIdx.x;
    const int pose = free_pose[hp];
    SGX_THREADS_BEGIN(tid)
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0;
    const SgxSE3 Tp = T[pose];
    for (int q = pose_start[pose] + tid; q < pose_start[pose + 1]; q += SGX_BA_THREADS) {
        const int k = pose_edges[q];
        const SgxBaEdge e = E[k];
        if (e.flags & 2) continue;
        const int stereo = e.flags & 1;
        const double Xl[3] = { X[3 * (size_t)e.point], X[3 * (size_t)e.point + 1], X[3 * (size_t)e.point + 2] };
        double A[3][3], Bj[3][6];
        sgx_ba_jacobians(Tp, Xl, stereo, cam.fx, cam.fy, cam.bf, A, Bj);
        const double info = (double)e.info;
        const double er[3] = { err[3 * (size_t)k], err[3 * (size_t)k + 1], stereo ? err[3 * (size_t)k + 2] : 0.0 };
        double rho1 = 1.0;
        if (e.flags & 4) { double r0; sgx_huber(sgx_po_chi2(er, info, stereo), stereo ? dStereo : dMono, &r0, &rho1); }
        const double w = rho1 * info;
        const double om[3] = { -(info * er[0]) * rho1, -(info * er[1]) * rho1, -(info * er[2]) * rho1 };
#pragma unroll
        for (int a = 0; a < 6; a++) {
            acc[21 + a] += Bj[0][a] * om[0] + Bj[1][a] * om[1] + Bj[2][a] * om[2];
#pragma unroll
            for (int c = 0; c < 6; c++) if (c >= a) acc[a * 6 - (a * (a - 1)) / 2 + (c - a)] += Bj[0][a] * w * Bj[0][c] + Bj[1][a] * w * Bj[1][c] + Bj[2][a] * w * Bj[2][c];
        }
    }
#pragma unroll
    for (int i = 0; i < 27; i++) part[tid * 27 + i] = acc[i];
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid < 27 * 8) { const int c = tid >> 3, j = tid & 7; double s = 0; for (int l = 0; l < 32; l++) s += part[(j * 32 + l) * 27 + c]; part2[c * 8 + j] = s; }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid < 27) {
        double s = 0; for (int l = 0; l < 8; l++) s += part2[tid * 8 + l];
        if (tid >= 21) bp[6 * (size_t)hp + (tid - 21)] = s;
        else {      // unpack upper-triangular index -> (a, c)
            int a = 0, rem = tid; while (rem >= 6 - a) { rem -= 6 - a; a++; }
            const int c = a + rem;
            Hpp[(size_t)hp * 36 + 6 * a + c] = s; Hpp[(size_t)hp * 36 + 6 * c + a] = s;
        }
    }
    SGX_THREADS_END
}

// k_ba_schur_init: S = blockdiag(Hpp) + lambda*I (setLambda block_solver.hpp:564-589, then "_Hschur = _Hpp" :372-373), coef = 0
SGX_KERNEL(SGX_BA_THREADS) k_ba_schur_init(int nf, const double *Hpp, double lambda, double *S, double *coef)
{
    SGX_THREADS_BEGIN(tid)
    const int NP = 6 * nf;
    const size_t total = (size_t)NP * NP;
    for (size_t i = (size_t)blockIdx.x * SGX_BA_THREADS + tid; i < total; i += (size_t)gridDim.x * SGX_BA_THREADS) {
        const int r = (int)(i / NP), c = (int)(i % NP);
        double v = 0;
        if (r / 6 == c / 6) v = Hpp[(size_t)(r / 6) * 36 + 6 * (r % 6) + (c % 6)] + (r == c ? lambda : 0.0);
        S[i] = v;
    }
    for (int i = (int)blockIdx.x * SGX_BA_THREADS + tid; i < NP; i += (int)gridDim.x * SGX_BA_THREADS) coef[i] = 0;
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// Reduced camera system  S xp = bp - coef  (S symmetric positive definite, n = 6 * free poses, row-major, full storage).
// The reference factorises it with Eigen SimplicialLDLT (G/solvers/linear_solver_eigen.h:94-124); the solution is unique,
// so a Cholesky L L^T is used here.  *ok is cleared when a pivot is not positive (or NaN): the LM step is then rejected
// exactly like solve()==false (levenberg.cpp:126-127).
//   n <= 128 : k_chol_small_reg — one 256-thread workgroup, the working matrix in registers, finished columns in LDS (k_chol_small, the LDS-resident
//              first version, stays as a comparison tap).
//   n  > 128 : blocked right-looking factorisation on 32x32 tiles staged in LDS, per panel k:
//                k_chol_diag_wave (ONE wave, tile in registers: factor L_kk and its explicit inverse Linv_kk; k_chol_diag = workgroup / LDS version, emulator + tap)
//                k_chol_panel  (one workgroup per row tile below:  L_ik = A_ik Linv_kk^T   — a tile GEMM)
//                k_chol_update (one workgroup per lower-triangular tile pair inside the current 256-column outer panel: A_ij -= L_ik L_jk^T)
//              per outer panel: k_chol_update_wide (fp64 MFMA rank-256 update of everything right of the panel, 64 x 64 tiles)
//              then the backward pass: k_chol_back_step per diagonal block (all CUs), or k_chol_solve (one workgroup) via the tuning tap.
// ---------------------------------------------------------------------------------------------
#define SGX_NB 32            /* tile edge of the blocked factorisation.  Measured: 64 halves the launches and runs the 2000-keyframe BA 8 % faster, but its
                                diagonal-tile kernel (64 serial steps + a 64-long register inverse) makes LocalBA-sized systems 20 % slower */
#define SGX_NB_SHIFT 5
#define SGX_TB (SGX_NB / 16)  /* register block edge of the tile GEMMs (16 x 16 thread map) */
#define SGX_CHOL_SMALL 128

SGX_KERNEL(256) k_chol_small(int n, const double *S, const double *bp, const double *coef, double *x, int *ok)
{
    // L D L^T in "unscaled column" form: column j keeps u_ij = L_ij * d_j, so no square roots and no separate column-scaling phase; the right-hand side
    // rides along as an extra column (forward substitution folded into the factorisation).  One barrier per column for the factorisation, one per
    // column for the back substitution  x_j = (v_j - sum_{i>j} u_ij x_i) / d_j.  Rows are padded to an odd stride (bank-conflict-free column walks);
    // the trailing update maps the 256 threads as 16 x 16 over (row, column) residues — no integer divisions in the loops.
    constexpr int LD = SGX_CHOL_SMALL + 1;
    SGX_LDS double A[SGX_CHOL_SMALL * LD];
    SGX_LDS double v[SGX_CHOL_SMALL];
    const int NT = (int)blockDim.x;
    SGX_THREADS_BEGIN(tid)
    for (int r = tid >> 4; r < n; r += NT >> 4)
        for (int c = tid & 15; c < n; c += 16) A[r * LD + c] = S[(size_t)r * n + c];
    for (int i = tid; i < n; i += NT) v[i] = bp[i] - coef[i];
    SGX_THREADS_END
    SGX_SYNC();
    int jfail = n;
    for (int j = 0; j < n; j++) {
        const double d = A[j * LD + j];                  // final pivot: every update from the columns before j has been applied
        if (!(d > 0)) { jfail = j; break; }
        const double rd = 1.0 / d, vj = v[j];
        SGX_THREADS_BEGIN(tid)
        const int ty = tid >> 4, tx = tid & 15;
        for (int i = j + 1 + ty; i < n; i += NT >> 4) {
            const double f = A[i * LD + j] * rd;         // L_ij
            for (int c = j + 1 + tx; c <= i; c += 16) A[i * LD + c] -= f * A[c * LD + j];
            if (tx == 0) v[i] -= f * vj;
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    if (jfail < n) {
        SGX_THREADS_BEGIN(tid) if (tid == 0) *ok = 0; SGX_THREADS_END
        return;
    }
    for (int j = n - 1; j >= 0; j--) {
        const double xj = v[j] / A[j * LD + j];
        SGX_THREADS_BEGIN(tid)
        for (int i = tid; i < j; i += NT) v[i] -= A[j * LD + i] * xj;
        if (tid == 0) x[j] = xj;
        SGX_THREADS_END
        SGX_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// k_chol_small_reg: the same factorisation and solve (n <= 128, unscaled L D L^T, right-hand side riding along) with the working matrix in REGISTERS.
// Thread (ty, tx) of the 16 x 16 map owns A[ty + 16 a][tx + 16 c], a, c = 0..7 (64 doubles) and a copy of the right-hand side entries v[tx + 16 c].
// Column j: its owners (tx == j mod 16) publish the finished column, unscaled, as row j of LmT in LDS (+ v_j in slot 128) — one barrier — and every thread
// applies the rank-1 update to its own registers from 2 x 8 broadcast reads of that row.  The 8 x 16 column steps are unrolled over the column group jb so
// that every register index is static and the row / column groups already eliminated (a, c < jb; c > a) are pruned at compile time: 120 multiply-add
// statements in total instead of 64 per column.  The LDS-resident version (k_chol_small) spends ~1.8 us per column on dependent LDS read-modify-writes;
// LmT doubles as the factor for the back substitution, done 16 unknowns at a time: every thread solves the 16 x 16 triangle redundantly in registers
// (no barriers inside), then threads 0..j0-1 push the block into their pending right-hand sides — 8 barriers instead of 128.
// (Measured and dropped: two columns per barrier, every thread correcting the second published column itself — 3.30 vs 3.05 ms per LocalBA of 20 + 40
// keyframes: the barrier is not what a column step waits for, the extra reads and corrections cost more than it saves.)
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(256) k_chol_small_reg(int n, const double *S, const double *bp, const double *coef, double *x, int *ok)
{
    constexpr int N = SGX_CHOL_SMALL, LD = SGX_CHOL_SMALL + 1, G = SGX_CHOL_SMALL / 16;
    SGX_LDS double LmT[N * LD];                       // LmT[j * LD + i] = u_ij = L_ij d_j (i >= j);  LmT[j * LD + N] = y_j (forward-substituted right-hand side)
    SGX_LDS double rdiag[N];                          // 1 / d_j
    SGX_LDS double w[N];                              // pending right-hand sides of the back substitution, then the solution
    SGX_PRIV_DECL(double, a, G * G, 256);
    SGX_PRIV_DECL(double, vr, G, 256);
    SGX_THREADS_BEGIN(tid)
    SGX_PRIV_BIND(a, tid); SGX_PRIV_BIND(vr, tid);
    const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
    for (int ai = 0; ai < G; ai++)
#pragma unroll
        for (int ci = 0; ci < G; ci++) { const int i = ty + 16 * ai, c = tx + 16 * ci; a[G * ai + ci] = (i < n && c < n) ? S[(size_t)i * n + c] : 0.0; }
#pragma unroll
    for (int ci = 0; ci < G; ci++) { const int c = tx + 16 * ci; vr[ci] = c < n ? bp[c] - coef[c] : 0.0; }
    SGX_THREADS_END
    bool failed = false;
#pragma unroll
    for (int jb = 0; jb < G; jb++) {
        for (int jj = 0; jj < 16; jj++) {
            const int j = 16 * jb + jj;
            if (j >= n || failed) break;
            SGX_THREADS_BEGIN(tid)
            SGX_PRIV_BIND(a, tid); SGX_PRIV_BIND(vr, tid);
            const int ty = tid >> 4, tx = tid & 15;
            if (tx == jj) {
#pragma unroll
                for (int ai = jb; ai < G; ai++) LmT[j * LD + ty + 16 * ai] = a[G * ai + jb];
                if (ty == 0) LmT[j * LD + N] = vr[jb];
            }
            SGX_THREADS_END
            SGX_SYNC();
            const double d = LmT[j * LD + j];            // final pivot: every update from the columns before j has been applied
            if (!(d > 0)) { failed = true; break; }
            const double rd = 1.0 / d, fv = LmT[j * LD + N] * rd;
            SGX_THREADS_BEGIN(tid)
            SGX_PRIV_BIND(a, tid); SGX_PRIV_BIND(vr, tid);
            const int ty = tid >> 4, tx = tid & 15;
            if (tid == 0) rdiag[j] = rd;
            double f[G], u[G];
#pragma unroll
            for (int g = jb; g < G; g++) {
                const double ui = LmT[j * LD + ty + 16 * g], uc = LmT[j * LD + tx + 16 * g];
                f[g] = (ty + 16 * g > j) ? ui * rd : 0.0;      // L_ij for the rows below the pivot, 0 for finished rows (their update is an exact no-op)
                u[g] = (tx + 16 * g > j) ? uc : 0.0;
            }
#pragma unroll
            for (int ai = jb; ai < G; ai++)
#pragma unroll
                for (int ci = jb; ci <= ai; ci++) {
                    const double t = f[ai] * u[ci];
                    a[G * ai + ci] -= (ci < ai || tx <= ty) ? t : 0.0;      // lower triangle only (the diagonal groups hold both halves)
                }
#pragma unroll
            for (int ci = jb; ci < G; ci++) vr[ci] -= fv * u[ci];
            SGX_THREADS_END
        }
    }
    if (failed) {
        SGX_THREADS_BEGIN(tid) if (tid == 0) *ok = 0; SGX_THREADS_END
        return;
    }
    // ---- back substitution  x_j = (y_j - sum_{i>j} u_ij x_i) / d_j, 16 unknowns per barrier
    SGX_THREADS_BEGIN(tid)
    if (tid < n) w[tid] = LmT[tid * LD + N];
    SGX_THREADS_END
    SGX_SYNC();
    for (int jb = (n - 1) / 16; jb >= 0; jb--) {
        const int j0 = 16 * jb;
        SGX_THREADS_BEGIN(tid)
        double xs[16];
#pragma unroll
        for (int q = 15; q >= 0; q--) {                  // every thread solves the 16 x 16 triangle (broadcast reads, static indices)
            const int j = j0 + q;
            double sacc = j < n ? w[j] : 0.0;
#pragma unroll
            for (int r = q + 1; r < 16; r++) sacc -= ((j0 + r < n && j < n) ? LmT[j * LD + j0 + r] : 0.0) * xs[r];
            xs[q] = j < n ? sacc * rdiag[j] : 0.0;
        }
        if (tid < j0) {                                  // push the block into the rows above it
            double sacc = w[tid];
#pragma unroll
            for (int r = 0; r < 16; r++) sacc -= (j0 + r < n ? LmT[tid * LD + j0 + r] : 0.0) * xs[r];
            w[tid] = sacc;
        }
        if (tid < 16 && j0 + tid < n) {
            double v = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) v = (r == tid) ? xs[r] : v;
            x[j0 + tid] = v;
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
}

// factor the diagonal tile (lower part of S overwritten with L_kk) and store Linv_kk (32x32, row-major, zero-padded) in Linv[k0/NB]
SGX_KERNEL(256) k_chol_diag(int n, int k0, double *S, double *Linv, int *ok, const double *bp, const double *coef, double *x)
{
    // Factorised in the same unscaled L D L^T form as k_chol_small (one barrier per column, 16 x 16 thread map, no integer divisions in the loops);
    // the Cholesky factor the panel / update kernels expect is recovered at the end: L_ij = u_ij / sqrt(d_j), L_jj = sqrt(d_j).
    SGX_LDS double A[SGX_NB][SGX_NB + 1];
    SGX_LDS double X[SGX_NB][SGX_NB + 1];
    SGX_LDS double sd[SGX_NB], rsd[SGX_NB];
    const int nb = min(SGX_NB, n - k0);
    const int NT = (int)blockDim.x;
    SGX_THREADS_BEGIN(tid)
    if (k0 == 0) for (int i = tid; i < n; i += NT) x[i] = bp[i] - coef[i];            // right-hand side of the reduced system (first panel only)
    for (int t = tid; t < SGX_NB * SGX_NB; t += NT) {
        const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1);
        A[r][c] = (r < nb && c < nb) ? S[(size_t)(k0 + r) * n + k0 + c] : 0.0;
        X[r][c] = 0;
    }
    SGX_THREADS_END
    SGX_SYNC();
    int jfail = nb;
    for (int j = 0; j < nb; j++) {
        const double d = A[j][j];
        if (!(d > 0)) { jfail = j; break; }
        const double rd = 1.0 / d;
        SGX_THREADS_BEGIN(tid)
        const int ty = tid >> 4, tx = tid & 15;
        for (int i = j + 1 + ty; i < nb; i += NT >> 4) {
            const double f = A[i][j] * rd;
            for (int c = j + 1 + tx; c <= i; c += 16) A[i][c] -= f * A[c][j];
        }
        SGX_THREADS_END
        SGX_SYNC();
    }
    if (jfail < nb) {
        SGX_THREADS_BEGIN(tid) if (tid == 0) *ok = 0; SGX_THREADS_END
        return;
    }
    SGX_THREADS_BEGIN(tid)
    if (tid < SGX_NB) { const double r_ = tid < nb ? sqrt(A[tid][tid]) : 1.0; sd[tid] = r_; rsd[tid] = 1.0 / r_; }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < SGX_NB * SGX_NB; t += NT) {
        const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1);
        if (r < nb && c < r) A[r][c] = A[r][c] / sd[c];
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid < nb) A[tid][tid] = sd[tid];
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid < SGX_NB) {
        // column tid of L^-1 by forward substitution, the column in registers and every index static: entries above the diagonal come out as exact
        // zeros by themselves (x_q = 0 for q < tid), so no thread-dependent loop bounds; the tile reads are broadcasts the compiler can hoist.
        // (A run-time-bounded loop over LDS here cost ~20 us per tile: ~500 dependent LDS round trips.)
        double xc[SGX_NB];
#pragma unroll
        for (int r = 0; r < SGX_NB; r++) {
            double sacc = (r == tid) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < r; q++) sacc -= A[r][q] * xc[q];
            xc[r] = sacc * rsd[r];
        }
#pragma unroll
        for (int r = 0; r < SGX_NB; r++) X[r][tid] = (tid < nb && r < nb) ? xc[r] : 0.0;
    }
    SGX_THREADS_END
    SGX_SYNC();
    // forward substitution rides along with the factorisation (right-looking): y_k = Linv_kk x_k here, x_i -= L_ik y_k in k_chol_panel
    SGX_THREADS_BEGIN(tid)
    if (tid < nb) { double sacc = 0; for (int q = 0; q <= tid; q++) sacc += X[tid][q] * x[k0 + q]; sd[tid] = sacc; }      // sd reused as y_k
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid < nb) x[k0 + tid] = sd[tid];
    double *Lo = Linv + (size_t)(k0 / SGX_NB) * SGX_NB * SGX_NB;
    for (int t = tid; t < SGX_NB * SGX_NB; t += NT) {
        const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1);
        if (r < nb && c <= r) S[(size_t)(k0 + r) * n + k0 + c] = A[r][c];
        Lo[t] = X[r][c];
    }
    SGX_THREADS_END
}

#ifndef SGX_EMU
// ---------------------------------------------------------------------------------------------
// k_chol_diag_wave: k_chol_diag as ONE wave with the tile in registers (device only; the emulator keeps the LDS version, same arithmetic).
// Lane r < 32 holds row r of the tile (32 doubles) — the factorisation's column step j needs the pivot and u_cj = A[c][j] of every row c > j, which are
// lane c's register j: a v_readlane with a compile-time lane (the j and c loops are fully unrolled), i.e. a scalar operand of the multiply-add, no LDS
// and no barrier.  ~500 (readlane pair + FMA) triples for the factorisation and again for the explicit inverse (column t of L^-1 in lane t), against
// 32 barrier-separated LDS read-modify-write phases in the workgroup version: 23 -> ~10 us per tile, which is the sequential spine of every blocked solve.
// Same operations in the same order as k_chol_diag for L and L^-1 (bit-identical tiles); y_k comes from an in-lane forward substitution.
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ double sgx_readlane_f64(double v, int lane)
{ return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane)); }

// returns false when a pivot is not positive (nothing is stored then)
static __device__ __attribute__((noinline)) bool sgx_chol_diag_wave_body(int lane, int n, int k0, double *S, double *Linv, const double *bp, const double *coef, double *x)
{
    const int row = lane & 31;
    const int nb = min(SGX_NB, n - k0);
    if (k0 == 0 && bp) for (int i = lane; i < n; i += 64) x[i] = bp[i] - coef[i];    // right-hand side of the reduced system (first panel only; the envelope solver sets it up itself: bp == NULL)
    double a[SGX_NB];
    {
        const double *src = S + (size_t)(k0 + min(row, nb - 1)) * n + k0;
#pragma unroll
        for (int c = 0; c < SGX_NB; c++) a[c] = (row < nb && c < nb) ? src[min(c, nb - 1)] : 0.0;
    }
    // ---- unscaled L D L^T: a[j] of lane i becomes u_ij = L_ij d_j
#pragma unroll
    for (int j = 0; j < SGX_NB; j++) {
        if (j < nb) {
            const double d = sgx_readlane_f64(a[j], j);
            if (!(d > 0)) return false;
            const double rd = 1.0 / d;
            const double f = a[j] * rd;
#pragma unroll
            for (int c = j + 1; c < SGX_NB; c++) a[c] -= f * sgx_readlane_f64(a[j], c);
        }
        __builtin_amdgcn_sched_barrier(0);               // one column at a time: the readlanes of later columns stay behind (hoisted, their scalar results spill into VGPR lanes)
    }
    // ---- Cholesky factor: L_ij = u_ij / sqrt(d_j), L_jj = sqrt(d_j)
    double dv = 1.0;                                     // lane j: d_j
#pragma unroll
    for (int j = 0; j < SGX_NB; j++) dv = (row == j && j < nb) ? a[j] : dv;
    const double sdv = sqrt(dv), rsdv = 1.0 / sdv;
#pragma unroll
    for (int j = 0; j < SGX_NB; j++) {
        const double sdj = sgx_readlane_f64(sdv, j), rj = sgx_readlane_f64(rsdv, j);
        a[j] = (row == j) ? (j < nb ? sdj : 0.0) : sgx_div_by_recip(a[j], sdj, rj);
    }
    // ---- explicit inverse: lane t computes column t of L^-1 by forward substitution (same order as k_chol_diag)
    // (round 6: an LDS copy of the finished factor read back as broadcasts instead of the readlane pairs measured SLOWER: 16 -> 19 us per tile)
    double xc[SGX_NB];
#pragma unroll
    for (int r = 0; r < SGX_NB; r++) {
        double sacc = (r == row) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < r; q++) sacc -= sgx_readlane_f64(a[q], r) * xc[q];
        xc[r] = (r < nb && row < nb) ? sacc * sgx_readlane_f64(rsdv, r) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- y_k = L_kk^-1 x_k by forward substitution (lane r holds entry r); x_i -= L_ik y_k follows in k_chol_panel
    double b = (row < nb) ? x[k0 + row] : 0.0;
#pragma unroll
    for (int q = 0; q < SGX_NB; q++) {
        if (q < nb) {
            const double yq = sgx_readlane_f64(b, q) * sgx_readlane_f64(rsdv, q);        // entry q has received every update from the entries before it
            b = (row == q) ? yq : ((row > q) ? b - a[q] * yq : b);
        }
    }
    if (lane < nb) x[k0 + lane] = b;
    // ---- stores: the lower triangle of L_kk over S, L_kk^-1 (zero-padded 32 x 32, row-major) into its slot
    if (lane < nb) {
        double *dst = S + (size_t)(k0 + lane) * n + k0;
#pragma unroll
        for (int c = 0; c < SGX_NB; c++) if (c <= lane) dst[c] = a[c];
    }
    double *Lo = Linv + (size_t)(k0 / SGX_NB) * SGX_NB * SGX_NB;
    if (lane < SGX_NB) {
#pragma unroll
        for (int r = 0; r < SGX_NB; r++) Lo[r * SGX_NB + lane] = xc[r];
    }
    return true;
}

__global__ void __launch_bounds__(64) k_chol_diag_wave(int n, int k0, double *S, double *Linv, int *ok, const double *bp, const double *coef, double *x)
{
    const int lane = (int)threadIdx.x;
    if (!sgx_chol_diag_wave_body(lane, n, k0, S, Linv, bp, coef, x) && lane == 0) *ok = 0;
}
#endif

// C(NB x NB) = P * Q^T for two LDS tiles stored TRANSPOSED ([q][row], zero-padded): thread (ty, tx) of the 16 x 16 map owns the TB x TB block rows TB*ty.., cols TB*tx..;
// per q it reads TB + TB contiguous doubles and issues TB*TB FMAs (1 LDS read per FMA at TB = 2, 0.5 at TB = 4, instead of 2 for the one-output-per-thread form).
SGX_DEV void sgx_tile_gemm_nt(const double (*PT)[SGX_NB + 4], const double (*QT)[SGX_NB + 4], int ty, int tx, double acc[SGX_TB][SGX_TB])
{
#pragma unroll
    for (int i = 0; i < SGX_TB; i++)
#pragma unroll
        for (int j = 0; j < SGX_TB; j++) acc[i][j] = 0;
#pragma unroll 8
    for (int q = 0; q < SGX_NB; q++) {
        double a[SGX_TB], b[SGX_TB];
#pragma unroll
        for (int i = 0; i < SGX_TB; i++) { a[i] = PT[q][SGX_TB * ty + i]; b[i] = QT[q][SGX_TB * tx + i]; }
#pragma unroll
        for (int i = 0; i < SGX_TB; i++)
#pragma unroll
            for (int j = 0; j < SGX_TB; j++) acc[i][j] += a[i] * b[j];
    }
}

// L_ik = A_ik Linv_kk^T for the row tile i = k0/NB + 1 + blockIdx.x, then the forward-substitution update x_i -= L_ik y_k
SGX_KERNEL(256) k_chol_panel(int n, int k0, double *S, const double *Linv, const int *ok, double *x)
{
    SGX_LDS double LiT[SGX_NB][SGX_NB + 4];      // Linv_kk transposed: [q][c]
    SGX_LDS double AT[SGX_NB][SGX_NB + 4];       // A_ik transposed: [q][r]
    SGX_LDS double LoT[SGX_NB][SGX_NB + 4];      // L_ik transposed: [c][r]
    if (!*ok) return;
    const int nb = min(SGX_NB, n - k0);
    const int r0 = k0 + SGX_NB * (1 + (int)blockIdx.x);
    const int nr = min(SGX_NB, n - r0);
    const int NT = (int)blockDim.x;
    const double *Lk = Linv + (size_t)(k0 / SGX_NB) * SGX_NB * SGX_NB;
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < SGX_NB * SGX_NB; t += NT) {
        const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1);
        LiT[c][r] = Lk[t];                                                            // Linv[r][c] (zero above the diagonal and past nb)
        AT[c][r] = (r < nr && c < nb) ? S[(size_t)(r0 + r) * n + k0 + c] : 0.0;
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int ty = tid >> 4, tx = tid & 15;
    double acc[SGX_TB][SGX_TB];
    sgx_tile_gemm_nt(AT, LiT, ty, tx, acc);                                            // out[r][c] = sum_q A[r][q] Linv[c][q]
#pragma unroll
    for (int i = 0; i < SGX_TB; i++)
#pragma unroll
        for (int j = 0; j < SGX_TB; j++) {
            const int r = SGX_TB * ty + i, c = SGX_TB * tx + j;
            LoT[c][r] = acc[i][j];                                                     // L_ik (transposed) for the substitution below
            if (r < nr && c < nb) S[(size_t)(r0 + r) * n + k0 + c] = acc[i][j];
        }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid < nr) { double vv = x[r0 + tid]; for (int q = 0; q < nb; q++) vv -= LoT[q][tid] * x[k0 + q]; x[r0 + tid] = vv; }      // forward substitution: x_i -= L_ik y_k
    SGX_THREADS_END
}

// A_ij -= L_ik L_jk^T for the lower-triangular tile pairs (i >= j) of the trailing matrix, column tiles j < gridDim.y only (the rest of the current
// 256-column outer panel; everything right of the panel is updated once per panel by k_chol_update_wide).  grid = (row tiles, column tiles).
SGX_KERNEL(256) k_chol_update(int n, int k0, double *S, const int *ok)
{
    SGX_LDS double LiT[SGX_NB][SGX_NB + 4];
    SGX_LDS double LjT[SGX_NB][SGX_NB + 4];
    if (!*ok) return;
    const int nb = min(SGX_NB, n - k0);
    const int bi = (int)blockIdx.x, bj = (int)blockIdx.y;
    if (bi < bj) return;
    const int r0 = k0 + SGX_NB * (1 + bi), c0 = k0 + SGX_NB * (1 + bj);
    const int nr = min(SGX_NB, n - r0), nc = min(SGX_NB, n - c0);
    const int NT = (int)blockDim.x;
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < SGX_NB * SGX_NB; t += NT) {
        const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1);
        LiT[c][r] = (r < nr && c < nb) ? S[(size_t)(r0 + r) * n + k0 + c] : 0.0;
        LjT[c][r] = (r < nc && c < nb) ? S[(size_t)(c0 + r) * n + k0 + c] : 0.0;
    }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    const int ty = tid >> 4, tx = tid & 15;
    double acc[SGX_TB][SGX_TB];
    sgx_tile_gemm_nt(LiT, LjT, ty, tx, acc);
#pragma unroll
    for (int i = 0; i < SGX_TB; i++)
#pragma unroll
        for (int j = 0; j < SGX_TB; j++) {
            const int r = SGX_TB * ty + i, c = SGX_TB * tx + j;
            if (r < nr && c < nc && !(bi == bj && c > r)) S[(size_t)(r0 + r) * n + c0 + c] -= acc[i][j];
        }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_chol_update_wide: the rank-PW update of everything right of a finished outer panel (columns p0 .. p0+pw of L, pw <= SGX_OB):
//   A[r][c] -= sum_{k in panel} L[r][k] L[c][k]      for the lower-triangular 64 x 64 tile pairs (r-tile >= c-tile) starting at row / column q0 = p0 + pw.
// This is the one dense fp64 contraction of the solver (n^3/3 multiply-adds for an n x n system) and runs on the matrix cores:
// v_mfma_f64_16x16x4_f64, A operand lane l = L[r0 + (l&15)][k + (l>>4)], B operand lane l = L[c0 + (l&15)][k + (l>>4)], C/D lane l, register i =
// row (l>>4) + 4 i, column l&15 (cdna guide: the f64 shape has its own C/D map).  A workgroup = 4 waves = one 64 x 64 tile, each wave a 32 x 32 quadrant
// (2 x 2 MFMA blocks, 16 accumulator doubles per lane); the two 64 x 32 operand slabs of a K-chunk are staged TRANSPOSED in LDS with a row stride of
// 80 doubles, which spreads the 16 rows x 4 k of an operand read over all 64 banks.  With rank-32 updates the trailing matrix made a round trip
// through L2 / HBM per 32 columns (560 GB per factorisation at n = 12 000, 0.74 s of 1.08 s); here it makes one per 256 columns and the
// multiply-adds leave the VALU.  grid = (tiles, tiles), upper-triangle workgroups exit.
// ---------------------------------------------------------------------------------------------
#define SGX_OB 256           /* outer panel width of the two-level blocked factorisation (8 tiles of SGX_NB) */
#define SGX_WT 64            /* output tile edge of the wide update */
#define SGX_WLD 80           /* LDS row stride (doubles) of the transposed operand slabs */
#ifndef SGX_EMU
typedef double sgx_f64x4 __attribute__((ext_vector_type(4)));
#endif
SGX_KERNEL(256) k_chol_update_wide(int n, int p0, int pw, double *S, const int *ok)
{
    if (!*ok) return;
    const int bi = (int)blockIdx.x, bj = (int)blockIdx.y;
    if (bi < bj) return;
    const int q0 = p0 + pw;
    const int r0 = q0 + SGX_WT * bi, c0 = q0 + SGX_WT * bj;
#ifndef SGX_EMU
    SGX_LDS double LiT[SGX_NB][SGX_WLD];          // [k][row] of the row tile
    SGX_LDS double LjT[SGX_NB][SGX_WLD];          // [k][row] of the column tile
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    sgx_f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = sgx_f64x4{0.0, 0.0, 0.0, 0.0};
    // a thread stages 8 consecutive k of one row (64 B) of each slab, 4 threads cover a row's K-chunk; the next chunk is fetched into registers
    // while the matrix cores work on the current one
    const int srow = tid >> 2, skq = (tid & 3) * 8;
    const bool vi = r0 + srow < n, vj = c0 + srow < n;                 // rows past the matrix edge: fetched from the last row (no branch around the loads), zeroed on the way to LDS
    const double *gi = S + (size_t)min(r0 + srow, n - 1) * n + p0 + skq, *gj = S + (size_t)min(c0 + srow, n - 1) * n + p0 + skq;
    double pi[8], pj[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { pi[u] = gi[u]; pj[u] = gj[u]; }
    for (int kc = 0; kc < pw; kc += SGX_NB) {                          // pw is a multiple of SGX_NB (whole outer panels only)
#pragma unroll
        for (int u = 0; u < 8; u++) { LiT[skq + u][srow] = vi ? pi[u] : 0.0; LjT[skq + u][srow] = vj ? pj[u] : 0.0; }
        __syncthreads();
        if (kc + SGX_NB < pw) {
#pragma unroll
            for (int u = 0; u < 8; u++) { pi[u] = gi[kc + SGX_NB + u]; pj[u] = gj[kc + SGX_NB + u]; }
        }
#pragma unroll
        for (int k4 = 0; k4 < SGX_NB; k4 += 4) {
            const int kk = k4 + (lane >> 4), rr = lane & 15;
            const double a0 = LiT[kk][wm + rr], a1 = LiT[kk][wm + 16 + rr];
            const double b0 = LjT[kk][wn + rr], b1 = LjT[kk][wn + 16 + rr];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = r0 + wm + 16 * a + (lane >> 4) + 4 * i, c = c0 + wn + 16 * b + (lane & 15);
                if (r < n && c < n && c <= r) S[(size_t)r * n + c] -= acc[a][b][i];
            }
#else
    // kernel-logic emulator: the same tile decomposition with a plain k-ordered sum (the matrix core's internal order differs in the last bits)
    SGX_THREADS_BEGIN(tid)
    for (int t = tid; t < SGX_WT * SGX_WT; t += 256) {
        const int r = r0 + (t >> 6), c = c0 + (t & 63);
        if (r < n && c < n && c <= r) {
            double sacc = 0;
            for (int k = 0; k < pw; k++) sacc += S[(size_t)r * n + p0 + k] * S[(size_t)c * n + p0 + k];
            S[(size_t)r * n + c] -= sacc;
        }
    }
    SGX_THREADS_END
#endif
}

// ---------------------------------------------------------------------------------------------
// Envelope (skyline) factorisation for SPARSE reduced camera systems.  The reference hands the reduced system to Eigen's SimplicialLDLT
// (G/solvers/linear_solver_eigen.h:94-124): what it costs follows the covisibility structure, not n^3.  With keyframes ordered along the trajectory the
// system is banded (a landmark is seen from a few consecutive keyframes; loop closures add a corner): on 32 x 32 tiles, tile row r is non-zero from its
// first tile column ft[r] on, fill stays inside that envelope, and column step k of the right-looking factorisation only touches the row tiles
// R(k) = { r > k : ft[r] <= k } (host-built lists).  For a narrow envelope the work per step is a handful of tile products, so the three launches per
// step of the dense path (diagonal tile, panel, update: ~45 us of launch latency and ramp per step, 375 steps at 12 000 unknowns) ARE the run time.
// k_chol_env_factor runs the WHOLE factorisation (and the forward substitution) as ONE persistent workgroup: per step, wave 0 factors and inverts the diagonal tile
// in registers (sgx_chol_diag_wave_body), then the 256-thread groups of the workgroup take the panel tiles L_rk = A_rk Linv_kk^T (+ x_r -= L_rk y_k) and the
// update pairs A_rc -= L_rk L_ck^T, r >= c in R(k), with workgroup barriers in between; k_chol_env_back is the backward pass in the same shape.
// The diagonal tile has the arithmetic of k_chol_diag (bit-identical); the tile products of the panel and the updates run on the fp64 matrix cores since round 6 (sgx_wave_gemm_nt_mfma:
// the matrix core's own summation order, last-bit differences against the emulator's k-ascending sums — as the dense path's k_chol_update_wide).
// The dense two-level path remains for systems whose envelope is not narrow.
// ---------------------------------------------------------------------------------------------
#define SGX_ENV_THREADS 512                       /* 8 waves: two per SIMD, so the diagonal-tile wave keeps its 32 x 32 tile + inverse in registers (no scratch) */
#define SGX_ENV_GROUPS (SGX_ENV_THREADS / 64)     /* one wave per tile product */
#define SGX_ENV_MAXM 8                            /* row tiles of one column step: one per wave, kept in LDS (8 x 9 KB + the inverse tile) */
// C(32 x 32) = P Q^T by ONE wave from two LDS tiles stored transposed ([q][row], stride NB + 4): lane (ty = lane >> 3, tx = lane & 7) owns the 4 x 4 block
// rows 4 ty.., columns 4 tx..; k ascending, one accumulation chain per entry (the order of sgx_tile_gemm_nt)
SGX_DEV void sgx_wave_gemm_nt(const double (*PT)[SGX_NB + 4], const double (*QT)[SGX_NB + 4], int ty, int tx, double acc[16])
{
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0;
#pragma unroll 4
    for (int q = 0; q < SGX_NB; q++) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = PT[q][4 * ty + i]; b[i] = QT[q][4 * tx + i]; }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[4 * i + j] += a[i] * b[j];
    }
}

#ifndef SGX_EMU
// the same product on the fp64 matrix cores (round 6): 2 x 2 tiles of v_mfma_f64_16x16x4f64, eight k-steps each — 32 MFMAs instead of 512 double FMAs per lane.  Operands straight
// from the transposed LDS tiles ([k][row]: lane (k = lane >> 4, row = lane & 15) of a k-step).  acc[a][b][i] = entry (16 a + 4 i + (lane >> 4), 16 b + (lane & 15)).
// The matrix core sums k in its own order: last-bit differences against sgx_wave_gemm_nt (as the dense path's k_chol_update_wide against its emulator form).
typedef double sgx_env_f64x4 __attribute__((ext_vector_type(4)));
SGX_DEV void sgx_wave_gemm_nt_mfma(const double (*PT)[SGX_NB + 4], const double (*QT)[SGX_NB + 4], int lane, sgx_env_f64x4 (&acc)[2][2])
{
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = sgx_env_f64x4{0.0, 0.0, 0.0, 0.0};
    const int kq = lane >> 4, rr = lane & 15;
#pragma unroll
    for (int k4 = 0; k4 < SGX_NB; k4 += 4) {
        const double a0 = PT[k4 + kq][rr], a1 = PT[k4 + kq][16 + rr], b0 = QT[k4 + kq][rr], b1 = QT[k4 + kq][16 + rr];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
}
#endif

#ifdef SGX_EMU
// the workgroup / LDS form of the diagonal-tile factorisation (k_chol_diag) for tile kd: what sgx_chol_diag_wave_body does on the device, same arithmetic.  Sets *fail on a non-positive pivot.
static void sgx_env_diag_emu(int kd, int n, double *S, double *Linv, const double *bp, const double *coef, double *x, double (*A)[SGX_NB + 1], double (*X)[SGX_NB + 1], double *sd, double *rsd, int *fail)
{
    const int k0 = kd * SGX_NB, nb = min(SGX_NB, n - k0);
    SGX_THREADS_BEGIN(tid)
    if (tid < 256) {
        if (k0 == 0 && bp) for (int i = tid; i < n; i += 256) x[i] = bp[i] - coef[i];
        for (int t = tid; t < SGX_NB * SGX_NB; t += 256) { const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1); A[r][c] = (r < nb && c < nb) ? S[(size_t)(k0 + r) * n + k0 + c] : 0.0; X[r][c] = 0; }
    }
    SGX_THREADS_END
    int jfail = nb;
    for (int j = 0; j < nb; j++) {
        const double d = A[j][j];
        if (!(d > 0)) { jfail = j; break; }
        const double rd = 1.0 / d;
        SGX_THREADS_BEGIN(tid)
        if (tid < 256) {
            const int ty = tid >> 4, tx = tid & 15;
            for (int i = j + 1 + ty; i < nb; i += 16) { const double f = A[i][j] * rd; for (int c = j + 1 + tx; c <= i; c += 16) A[i][c] -= f * A[c][j]; }
        }
        SGX_THREADS_END
    }
    if (jfail < nb) { *fail = 1; return; }
    SGX_THREADS_BEGIN(tid)
    if (tid < SGX_NB) { const double r_ = tid < nb ? sqrt(A[tid][tid]) : 1.0; sd[tid] = r_; rsd[tid] = 1.0 / r_; }
    SGX_THREADS_END
    SGX_THREADS_BEGIN(tid)
    if (tid < 256) for (int t = tid; t < SGX_NB * SGX_NB; t += 256) { const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1); if (r < nb && c < r) A[r][c] = A[r][c] / sd[c]; }
    SGX_THREADS_END
    SGX_THREADS_BEGIN(tid)
    if (tid < nb) A[tid][tid] = sd[tid];
    SGX_THREADS_END
    SGX_THREADS_BEGIN(tid)
    if (tid < SGX_NB) {
        double xc[SGX_NB];
        for (int r = 0; r < SGX_NB; r++) { double sacc = (r == tid) ? 1.0 : 0.0; for (int q = 0; q < r; q++) sacc -= A[r][q] * xc[q]; xc[r] = sacc * rsd[r]; }
        for (int r = 0; r < SGX_NB; r++) X[r][tid] = (tid < nb && r < nb) ? xc[r] : 0.0;
    }
    SGX_THREADS_END
    SGX_THREADS_BEGIN(tid)
    if (tid < nb) { double sacc = 0; for (int q = 0; q <= tid; q++) sacc += X[tid][q] * x[k0 + q]; sd[tid] = sacc; }
    SGX_THREADS_END
    SGX_THREADS_BEGIN(tid)
    if (tid < nb) x[k0 + tid] = sd[tid];
    if (tid < 256) {
        double *Lo = Linv + (size_t)kd * SGX_NB * SGX_NB;
        for (int t = tid; t < SGX_NB * SGX_NB; t += 256) { const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1); if (r < nb && c <= r) S[(size_t)(k0 + r) * n + k0 + c] = A[r][c]; Lo[t] = X[r][c]; }
    }
    SGX_THREADS_END
}
#endif

// update pair p of column step k: A_rc -= L_rk L_ck^T with (r, c) = the p-th pair (bi >= bj, row-major over the lower triangle) of R(k), both operands in the LDS pool
// sep0 < n: this workgroup is the SECOND branch of a two-branch elimination — its contributions to the separator x separator tiles go to the side buffer S2 (nsep x nsep),
// which the separator phase adds to S (the first branch subtracts in S itself; the two never write the same word)
SGX_DEV void sgx_env_update_pair(int pidx, int lane, int n, int q0, const int *rows, const double (*pool)[SGX_NB][SGX_NB + 4], double *S, int sep0, double *S2)
{
    int bi = (int)((sqrt(8.0 * pidx + 1.0) - 1.0) * 0.5); while ((bi + 1) * (bi + 2) / 2 <= pidx) bi++; while (bi * (bi + 1) / 2 > pidx) bi--;
    const int bj = pidx - bi * (bi + 1) / 2;
    const int r0 = rows[q0 + bi] * SGX_NB, c0 = rows[q0 + bj] * SGX_NB, nr = min(SGX_NB, n - r0), nc = min(SGX_NB, n - c0);
#ifndef SGX_EMU
    sgx_env_f64x4 u4[2][2];
    sgx_wave_gemm_nt_mfma(pool[bi], pool[bj], lane, u4);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 16 * a + 4 * i + (lane >> 4), c = 16 * b + (lane & 15);
                if (r < nr && c < nc && !(bi == bj && c > r)) {
                    if (c0 >= sep0) S2[(size_t)(r0 - sep0 + r) * (n - sep0) + (c0 - sep0 + c)] -= u4[a][b][i];
                    else S[(size_t)(r0 + r) * n + c0 + c] -= u4[a][b][i];
                }
            }
#else
    const int ty = lane >> 3, tx = lane & 7;
    double u[16];
    sgx_wave_gemm_nt(pool[bi], pool[bj], ty, tx, u);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = 4 * ty + i, c = 4 * tx + j;
            if (r < nr && c < nc && !(bi == bj && c > r)) {
                if (c0 >= sep0) S2[(size_t)(r0 - sep0 + r) * (n - sep0) + (c0 - sep0 + c)] -= u[4 * i + j];
                else S[(size_t)(r0 + r) * n + c0 + c] -= u[4 * i + j];
            }
        }
#endif
}

#ifndef SGX_EMU
// The update pairs of one wave, software-pipelined (round 6): a pair is a read-modify-write of one 32 x 32 tile of S in global memory around a 0.5 us matrix-core product — done one
// after the other, the load latency of every tile (1-3 us, the band is larger than L2) was the step time of waves 1-7.  The target tile of the NEXT pair is fetched into
// registers before the current product starts.  Same arithmetic and the same tile -> wave assignment as a loop over sgx_env_update_pair.
struct SgxEnvUpd { int bi, bj, nr, nc; double *base; size_t ld; };
SGX_DEV void sgx_env_upd_target(int pidx, int n, int q0, const int *rows, double *S, int sep0, double *S2, SgxEnvUpd &u)
{
    int bi = (int)((sqrt(8.0 * pidx + 1.0) - 1.0) * 0.5); while ((bi + 1) * (bi + 2) / 2 <= pidx) bi++; while (bi * (bi + 1) / 2 > pidx) bi--;
    const int bj = pidx - bi * (bi + 1) / 2;
    const int r0 = rows[q0 + bi] * SGX_NB, c0 = rows[q0 + bj] * SGX_NB;
    u.bi = bi; u.bj = bj; u.nr = min(SGX_NB, n - r0); u.nc = min(SGX_NB, n - c0);
    if (c0 >= sep0) { u.ld = (size_t)(n - sep0); u.base = S2 + (size_t)(r0 - sep0) * u.ld + (c0 - sep0); }
    else { u.ld = (size_t)n; u.base = S + (size_t)r0 * n + c0; }
}
SGX_DEV bool sgx_env_upd_mine(const SgxEnvUpd &u, int a, int b, int i, int lane, int &r, int &c)
{
    r = 16 * a + 4 * i + (lane >> 4); c = 16 * b + (lane & 15);
    return r < u.nr && c < u.nc && !(u.bi == u.bj && c > r);
}
SGX_DEV void sgx_env_upd_load(const SgxEnvUpd &u, int lane, sgx_env_f64x4 (&v)[2][2])
{
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++) { int r, c; v[a][b][i] = sgx_env_upd_mine(u, a, b, i, lane, r, c) ? u.base[(size_t)r * u.ld + c] : 0.0; }
}
SGX_DEV void sgx_env_upd_store(const SgxEnvUpd &u, int lane, const sgx_env_f64x4 (&v)[2][2], const sgx_env_f64x4 (&p)[2][2])
{
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 4; i++) { int r, c; if (sgx_env_upd_mine(u, a, b, i, lane, r, c)) u.base[(size_t)r * u.ld + c] = v[a][b][i] - p[a][b][i]; }
}
SGX_DEV void sgx_env_update_pairs(int first, int stride, int npairs, int lane, int n, int q0, const int *rows, const double (*pool)[SGX_NB][SGX_NB + 4], double *S, int sep0, double *S2)
{
    if (first >= npairs) return;
    SgxEnvUpd uc, un; sgx_env_f64x4 cur[2][2], nxt[2][2], prod[2][2];
    sgx_env_upd_target(first, n, q0, rows, S, sep0, S2, uc); sgx_env_upd_load(uc, lane, cur);
    for (int p = first; p < npairs; p += stride) {
        const bool more = p + stride < npairs;
        if (more) { sgx_env_upd_target(p + stride, n, q0, rows, S, sep0, S2, un); sgx_env_upd_load(un, lane, nxt); }
        sgx_wave_gemm_nt_mfma(pool[uc.bi], pool[uc.bj], lane, prod);
        sgx_env_upd_store(uc, lane, cur, prod);
        if (more) {
            uc = un;
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) cur[a][b] = nxt[a][b];
        }
    }
}
#endif

// Schedule.  The diagonal tile is the sequential spine (about 10 us on one wave); the updates of a step are up to 36 tile products.  Both run at the same time:
// after the panel of step k, wave 0 takes the FIRST update pair — rows ascend inside a step, so that is tile (k + 1, k + 1) whenever row k + 1 belongs to R(k) — and then
// factors and inverts diagonal tile k + 1 (it has received every update: the earlier steps' before their closing barrier, step k's just now, by the same wave), while
// waves 1 .. 7 work through the remaining pairs.  One barrier later panel k + 1 finds Linv_{k+1} and all of column k + 1 ready.
//
// Two branches.  A band can be eliminated from both ends at once: the host orders the unknowns [first half ascending][second half DEScending][separator], so that the two
// halves are banded blocks that only meet in the separator's rows; their column steps are independent (no tile of one is touched by the other) and run as two workgroups
// of the same launch (phase 0: workgroup 0 takes steps [0, nA), workgroup 1 takes [nA, nA + nB)).  Both update separator x separator tiles and the separator's share of
// the right-hand side: workgroup 0 in place, workgroup 1 in S2 / x2 (zeroed by the host), which the separator launch (phase 1: steps [nA + nB, nt), one workgroup) adds in
// before it starts.  nB = 0: everything is phase 0 of a single workgroup.  The sequential spine is half as long.
SGX_KERNEL(SGX_ENV_THREADS) k_chol_env_factor(int n, int nt, const int *rstart, const int *rows, double *S, double *Linv, int *ok, const double *bp, const double *coef, double *x, int dbg,
                                              int phase, int nA, int nB, double *S2, double *x2)
{
    const int branch = phase == 0 ? (int)blockIdx.x : 2;                 // 0 / 1: the two independent halves, 2: the separator
    const int kbeg = branch == 0 ? 0 : (branch == 1 ? nA : nA + nB), kend = branch == 0 ? nA : (branch == 1 ? nA + nB : nt);
    const int sepu = min(n, (nA + nB) * SGX_NB);                         // first unknown of the separator
    const int sep0 = branch == 1 ? sepu : n;                             // workgroup 1 diverts its separator contributions
    double *xs = branch == 1 ? x2 - sepu : x;                            // ... and its share of the separator's right-hand side
    SGX_LDS double LiT[SGX_NB][SGX_NB + 4];                      // Linv_kk transposed: [q][c]
    SGX_LDS double yk[SGX_NB];
    SGX_LDS double pool[SGX_ENV_MAXM][SGX_NB][SGX_NB + 4];       // the row tiles of the step, transposed [q][r]: A_rk on the way in, L_rk on the way out
    SGX_LDS int s_fail;
    SGX_PRIV_DECL(double, acc, 16, SGX_ENV_THREADS);
#ifdef SGX_EMU
    SGX_LDS double A[SGX_NB][SGX_NB + 1];
    SGX_LDS double X[SGX_NB][SGX_NB + 1];
    SGX_LDS double sd[SGX_NB], rsd[SGX_NB];
#endif
    if (kbeg >= kend) return;
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) s_fail = 0;
    // right-hand side x = bp - coef of the steps this workgroup owns (workgroup 0 also the separator's); the separator launch folds workgroup 1's contributions in
    if (branch == 0) { for (int i = tid; i < nA * SGX_NB && i < n; i += SGX_ENV_THREADS) x[i] = bp[i] - coef[i]; for (int i = sepu + tid; i < n; i += SGX_ENV_THREADS) x[i] = bp[i] - coef[i]; }
    else if (branch == 1) { for (int i = nA * SGX_NB + tid; i < sepu; i += SGX_ENV_THREADS) x[i] = bp[i] - coef[i]; }
    else if (nB > 0) {
        const int ns = n - sepu;
        for (int i = tid; i < ns; i += SGX_ENV_THREADS) x[sepu + i] += x2[i];
        for (int t = tid; t < ns * ns; t += SGX_ENV_THREADS) { const int r = t / ns, c = t - r * ns; if (c <= (r | (SGX_NB - 1))) S[(size_t)(sepu + r) * n + sepu + c] += S2[t]; }     // lower tiles (the diagonal tiles whole: their upper halves are never read)
    }
    SGX_THREADS_END
    SGX_SYNC();
    // ---- first diagonal tile of the range: L_kk over S, Linv_kk, y_k = Linv_kk x_k
#ifndef SGX_EMU
    if ((int)threadIdx.x < 64 && !(dbg & 1)) { if (!sgx_chol_diag_wave_body((int)threadIdx.x, n, kbeg * SGX_NB, S, Linv, nullptr, coef, x) && threadIdx.x == 0) s_fail = 1; }
#else
    sgx_env_diag_emu(kbeg, n, S, Linv, nullptr, coef, x, A, X, sd, rsd, &s_fail);
#endif
    SGX_SYNC();
    for (int k = kbeg; k < kend; k++) {
        if (s_fail) {
            SGX_THREADS_BEGIN(tid) if (tid == 0) *ok = 0; SGX_THREADS_END
            return;
        }
        const int k0 = k * SGX_NB, nb = min(SGX_NB, n - k0);
        const int q0 = rstart[k], m = (dbg & 2) ? 0 : rstart[k + 1] - q0;   // m <= SGX_ENV_MAXM (host)
        if (m > 0) {
            const double *Lk = Linv + (size_t)k * SGX_NB * SGX_NB;
            // ---- panel: L_rk = A_rk Linv_kk^T for r in R(k) (wave g takes row tile g), then x_r -= L_rk y_k
#ifndef SGX_EMU
            {   // a row tile belongs to ONE wave from the load to the forward substitution: only Linv_kk / y_k are shared, so one workgroup barrier (after staging) and wave-level
                // ordering inside (the emulator keeps the four barrier-separated phases below: it runs the threads of a phase one after the other)
                const int tid = (int)threadIdx.x, g = tid >> 6, lane = tid & 63;
                for (int t = tid; t < SGX_NB * SGX_NB; t += SGX_ENV_THREADS) { const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1); LiT[c][r] = Lk[t]; }
                if (tid < SGX_NB) yk[tid] = tid < nb ? x[k0 + tid] : 0.0;                                          // y_k for the forward substitution
                const int r0 = g < m ? rows[q0 + g] * SGX_NB : 0, nr = min(SGX_NB, n - r0);
                if (g < m) {
                    for (int t = lane; t < SGX_NB * SGX_NB; t += 64) {
                        const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1);
                        pool[g][c][r] = (r < nr && c < nb) ? S[(size_t)(r0 + r) * n + k0 + c] : 0.0;
                    }
                }
                __syncthreads();
                if (g < m) {
                    sgx_env_f64x4 a4[2][2];
                    sgx_wave_gemm_nt_mfma(pool[g], LiT, lane, a4);                                                      // out[r][c] = sum_q A[r][q] Linv[c][q]
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();            // every lane has read A_rk before L_rk replaces it
#pragma unroll
                    for (int a = 0; a < 2; a++)
#pragma unroll
                        for (int b = 0; b < 2; b++)
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                const int r = 16 * a + 4 * i + (lane >> 4), c = 16 * b + (lane & 15);
                                pool[g][c][r] = a4[a][b][i];
                                if (r < nr && c < nb) S[(size_t)(r0 + r) * n + k0 + c] = a4[a][b][i];
                            }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
                    double *xr = r0 >= sep0 ? xs : x;                                                                   // second branch: separator rows accumulate in x2
                    if (lane < nr) { double vv = xr[r0 + lane]; for (int q = 0; q < nb; q++) vv -= pool[g][q][lane] * yk[q]; xr[r0 + lane] = vv; }      // forward substitution
                }
            }
#else
            SGX_THREADS_BEGIN(tid)
            for (int t = tid; t < SGX_NB * SGX_NB; t += SGX_ENV_THREADS) { const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1); LiT[c][r] = Lk[t]; }
            if (tid < SGX_NB) yk[tid] = tid < nb ? x[k0 + tid] : 0.0;                                              // y_k for the forward substitution
            const int g = tid >> 6, lane = tid & 63;
            if (g < m) {
                const int r0 = rows[q0 + g] * SGX_NB, nr = min(SGX_NB, n - r0);
                for (int t = lane; t < SGX_NB * SGX_NB; t += 64) {
                    const int r = t >> SGX_NB_SHIFT, c = t & (SGX_NB - 1);
                    pool[g][c][r] = (r < nr && c < nb) ? S[(size_t)(r0 + r) * n + k0 + c] : 0.0;
                }
            }
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            SGX_PRIV_BIND(acc, tid);
            const int g = tid >> 6, lane = tid & 63;
            if (g < m) sgx_wave_gemm_nt(pool[g], LiT, lane >> 3, lane & 7, acc);                                    // out[r][c] = sum_q A[r][q] Linv[c][q]
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            SGX_PRIV_BIND(acc, tid);
            const int g = tid >> 6, lane = tid & 63;
            if (g < m) {
                const int r0 = rows[q0 + g] * SGX_NB, nr = min(SGX_NB, n - r0), ty = lane >> 3, tx = lane & 7;
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int r = 4 * ty + i, c = 4 * tx + j;
                        pool[g][c][r] = acc[4 * i + j];                                                               // L_rk (transposed) replaces A_rk
                        if (r < nr && c < nb) S[(size_t)(r0 + r) * n + k0 + c] = acc[4 * i + j];
                    }
            }
            SGX_THREADS_END
            SGX_SYNC();
            SGX_THREADS_BEGIN(tid)
            const int g = tid >> 6, lane = tid & 63;
            if (g < m) {
                const int r0 = rows[q0 + g] * SGX_NB, nr = min(SGX_NB, n - r0);
                double *xr = r0 >= sep0 ? xs : x;                                                                        // second branch: separator rows accumulate in x2
                if (lane < nr) { double vv = xr[r0 + lane]; for (int q = 0; q < nb; q++) vv -= pool[g][q][lane] * yk[q]; xr[r0 + lane] = vv; }      // forward substitution
            }
            SGX_THREADS_END
#endif
            SGX_SYNC();                                                                                            // x_{k+1} is complete before wave 0 turns it into y_{k+1}
        }
        // ---- updates A_rc -= L_rk L_ck^T for the pairs r >= c of R(k), and — at the same time, on wave 0 — diagonal tile k + 1
        const int npairs = (dbg & 4) ? 0 : m * (m + 1) / 2;
#ifndef SGX_EMU
        {
            const int tid = (int)threadIdx.x, g = tid >> 6, lane = tid & 63;
            if (g == 0) {
                if (npairs > 0) { sgx_env_update_pair(0, lane, n, q0, rows, pool, S, sep0, S2); __threadfence_block(); }         // tile (k + 1, k + 1) when row k + 1 is in R(k): stored before it is read back
                if (k + 1 < kend && !(dbg & 1)) { if (!sgx_chol_diag_wave_body(lane, n, (k + 1) * SGX_NB, S, Linv, nullptr, coef, x) && lane == 0) s_fail = 1; }
            } else {
                sgx_env_update_pairs(g, SGX_ENV_GROUPS - 1, npairs, lane, n, q0, rows, pool, S, sep0, S2);
            }
        }
#else
        SGX_THREADS_BEGIN(tid)
        const int g = tid >> 6, lane = tid & 63;
        for (int pidx = g; pidx < npairs; pidx += SGX_ENV_GROUPS) sgx_env_update_pair(pidx, lane, n, q0, rows, pool, S, sep0, S2);
        SGX_THREADS_END
        if (k + 1 < kend) sgx_env_diag_emu(k + 1, n, S, Linv, nullptr, coef, x, A, X, sd, rsd, &s_fail);
#endif
        SGX_SYNC();
    }
    if (s_fail) { SGX_THREADS_BEGIN(tid) if (tid == 0) *ok = 0; SGX_THREADS_END }
}

// backward pass over the envelope, one persistent workgroup: for k = nt-1 .. 0:  x_k = Linv_kk^T (y_k - sum_{r in R(k)} L_rk^T x_r)
// (column-oriented form of k_chol_back_step's row updates: the contributions of the finished blocks below are gathered when block k is solved).
// Thread (row q = tid >> 5, column c = tid & 31) multiplies entry (q, c) of every tile of R(k) with x_r[q] (independent loads), the 32 partial sums of a
// column meet in LDS; Linv_kk is fetched while they are formed.
// Two-branch plans: phase 0 = the separator's steps (one workgroup), phase 1 = the two halves, one workgroup each (they only read the separator's finished solution).
SGX_KERNEL(1024) k_chol_env_back(int n, int nt, const int *rstart, const int *rows, const double *S, const double *Linv, const double *y, double *xsol, const int *ok, int phase, int nA, int nB)
{
    SGX_LDS double ys[SGX_NB];
    SGX_LDS double part[SGX_NB][SGX_NB + 1];
    SGX_LDS double Li[SGX_NB][SGX_NB + 1];
    if (!*ok) return;
    const int branch = phase == 0 ? 2 : (int)blockIdx.x;
    const int kbeg = branch == 0 ? 0 : (branch == 1 ? nA : nA + nB), kend = branch == 0 ? nA : (branch == 1 ? nA + nB : nt);
    for (int k = kend - 1; k >= kbeg; k--) {
        const int k0 = k * SGX_NB, nb = min(SGX_NB, n - k0);
        const int q0 = rstart[k], m = rstart[k + 1] - q0;
        const double *Lk = Linv + (size_t)k * SGX_NB * SGX_NB;
        SGX_THREADS_BEGIN(tid)
        const int c = tid & 31, q = tid >> 5;
        double acc = 0;
        if (c < nb)
            for (int i = 0; i < m; i++) {
                const int r0 = rows[q0 + i] * SGX_NB;
                if (r0 + q < n) acc += S[(size_t)(r0 + q) * n + k0 + c] * xsol[r0 + q];
            }
        part[q][c] = acc;
        Li[q][c] = Lk[tid];
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        if (tid < SGX_NB) { double t = tid < nb ? y[k0 + tid] : 0.0; for (int s2 = 0; s2 < SGX_NB; s2++) t -= part[s2][tid]; ys[tid] = t; }
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        if (tid < nb) { double sacc = 0; for (int q = tid; q < nb; q++) sacc += Li[q][tid] * ys[q]; xsol[k0 + tid] = sacc; }
        SGX_THREADS_END
        SGX_SYNC();
    }
}

// x = (bp - coef); L y = x; L^T x = y — blocked with the stored diagonal inverses, one 256-thread workgroup
SGX_KERNEL(256) k_chol_solve(int n, const double *S, const double *Linv, const double *bp, const double *coef, double *x, const int *ok)
{
    SGX_LDS double ys[SGX_NB];
    if (!*ok) return;
    const int NT = (int)blockDim.x;
    // x already holds y = L^-1 (bp - coef): the forward substitution ran inside k_chol_diag / k_chol_panel.  Backward pass:
    for (int k0 = ((n - 1) / SGX_NB) * SGX_NB; k0 >= 0; k0 -= SGX_NB) {
        const int nb = min(SGX_NB, n - k0);
        const double *Lk = Linv + (size_t)(k0 / SGX_NB) * SGX_NB * SGX_NB;
        SGX_THREADS_BEGIN(tid)
        if (tid < nb) { double sacc = 0; for (int q = tid; q < nb; q++) sacc += Lk[q * SGX_NB + tid] * x[k0 + q]; ys[tid] = sacc; }    // x_k = Linv_kk^T y_k
        SGX_THREADS_END
        SGX_SYNC();
        SGX_THREADS_BEGIN(tid)
        if (tid < nb) x[k0 + tid] = ys[tid];
        for (int i = tid; i < k0; i += NT) { double vv = x[i]; for (int q = 0; q < nb; q++) vv -= S[(size_t)(k0 + q) * n + i] * ys[q]; x[i] = vv; }
        SGX_THREADS_END
        SGX_SYNC();
    }
}

// One step of the same backward pass spread over the chip (large systems: the single workgroup above would stream the whole factor, n^2/2 doubles,
// through one CU).  Launched once per diagonal block, last block first: every workgroup recomputes x_k = Linv_kk^T y_k (a 32 x 32 mat-vec) from the
// working vector y, workgroup 0 stores it to `xsol`, and workgroup w applies the update y_i -= sum_q L[k0+q][i] x_k[q] to its 256 columns i < k0
// (32 coalesced row segments).  y[k0..] is only read during the launch and `xsol` only written, so the workgroups need no ordering among themselves.
SGX_KERNEL(256) k_chol_back_step(int n, int k0, const double *S, const double *Linv, double *y, double *xsol, const int *ok)
{
    SGX_LDS double ys[SGX_NB];
    if (!*ok) return;
    const int nb = min(SGX_NB, n - k0);
    const double *Lk = Linv + (size_t)(k0 / SGX_NB) * SGX_NB * SGX_NB;
    SGX_THREADS_BEGIN(tid)
    if (tid < nb) { double sacc = 0; for (int q = tid; q < nb; q++) sacc += Lk[q * SGX_NB + tid] * y[k0 + q]; ys[tid] = sacc; }
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (blockIdx.x == 0 && tid < nb) xsol[k0 + tid] = ys[tid];
    const int i = (int)blockIdx.x * 256 + tid;
    if (i < k0) {
        double vv = y[i];
#pragma unroll 8
        for (int q = 0; q < nb; q++) vv -= S[(size_t)(k0 + q) * n + i] * ys[q];
        y[i] = vv;
    }
    SGX_THREADS_END
}

// ---------------------------------------------------------------------------------------------
// k_ba_schur_pairs: the Schur complement (block_solver.hpp:380-433) as one thread per (job, entry): a job is a pair of
// active free-pose edges (k1, k2) of one landmark (host-built list, static during one optimize() call); the 36 threads of a
// job produce the 36 entries of  Hpl_1 Dinv Hpl_2^T,  which are subtracted from S(i1,i2); the diagonal jobs also add coef(i1) += Hpl_1 (Dinv bl).
// k_ba_dinv (one thread per landmark) computes Dinv = (Hll + lambda I)^-1 first (closed form = Eigen Matrix3d::inverse()).
// ---------------------------------------------------------------------------------------------
SGX_KERNEL(SGX_BA_THREADS) k_ba_dinv(int nl, const uint8_t *pt_active, const double *Hll, double lambda, double *Dinv)
{
    SGX_THREADS_BEGIN(tid)
    const int l = (int)blockIdx.x * SGX_BA_THREADS + tid;
    if (l < nl) {
        double Di[9];
        if (!pt_active[l]) {
#pragma unroll
            for (int i = 0; i < 9; i++) Di[i] = 0;
        } else {
            double M[9];
#pragma unroll
            for (int i = 0; i < 9; i++) M[i] = Hll[(size_t)l * 9 + i];
            M[0] += lambda; M[4] += lambda; M[8] += lambda;
            const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
            const double det = M[0] * c00 + M[1] * c01 + M[2] * c02, id = 1.0 / det;
            Di[0] = c00 * id; Di[1] = (M[2] * M[7] - M[1] * M[8]) * id; Di[2] = (M[1] * M[5] - M[2] * M[4]) * id;
            Di[3] = c01 * id; Di[4] = (M[0] * M[8] - M[2] * M[6]) * id; Di[5] = (M[2] * M[3] - M[0] * M[5]) * id;
            Di[6] = c02 * id; Di[7] = (M[1] * M[6] - M[0] * M[7]) * id; Di[8] = (M[0] * M[4] - M[1] * M[3]) * id;
        }
#pragma unroll
        for (int i = 0; i < 9; i++) Dinv[(size_t)l * 9 + i] = Di[i];
    }
    SGX_THREADS_END
}

struct SgxBaJob { int k1, k2; };

#ifndef SGX_EMU
// ---------------------------------------------------------------------------------------------
// Schur job list built on the device (round 6; the host builder — build_jobs_host in sgx_ba.cpp — stays as the emulator's path and as the A/B arm of the tap build).
// The list is: all ordered pairs (k1, k2) of active free-pose edges of one landmark, grouped by destination block (i1, i2) of the reduced system, block rows ascending, i2 ascending
// inside a row, and inside a block the order in which the reference subtracts them (block_solver.hpp:380-433): the edges k1 of pose i1 in ascending landmark order, for each the edges
// k2 of its landmark in edge order.  One wave per block row i1:
//   count  lanes over the row's edges, LDS counters per i2                                      -> jobs and non-empty blocks of the row
//   scan   (k_ba_jobs_scan, one workgroup) exclusive sums over the rows                          -> where the row's jobs / block starts begin; totals for the host
//   fill   the same counters, scanned over i2 into cursors; then the row's edges ONE AFTER THE OTHER (64 of them fetched at a time), lanes over the landmark's edges: a job goes
//          to cursor[i2] + (lower lanes with the same i2), the last lane of an i2 group advances the cursor — a stable counting sort without atomics, so the list, hence
//          the reduced system, is the same bits as the host-built one.
// An edge is active when its level is 0 (flags bit 1 clear: k_ba_classify sets it) and its pose is free (hidx >= 0).
// ---------------------------------------------------------------------------------------------
SGX_DEV int sgx_ba_edge_row(const SgxBaEdge *E, const int *hidx, int k) { const int fl = E[k].flags, p = E[k].pose; return (fl & 2) ? -1 : hidx[p]; }

template <int FILL>
__global__ void __launch_bounds__(64) k_ba_jobs_row(int nf, const int *free_pose, const int *pose_start, const int *pose_edges_l, const int *pt_start, const int *pt_edges,
                                                    const SgxBaEdge *E, const int *hidx, int *row_jobs, int *row_blks, const int *job_off, const int *blk_off,
                                                    SgxBaJob *jobs, int *blk_start)
{
    extern __shared__ int sgx_ba_cnt[];                           /* nf counters, then cursors */
    int *cnt = sgx_ba_cnt;
    const int h1 = (int)blockIdx.x, lane = (int)threadIdx.x, p = free_pose[h1];
    for (int i = lane; i < nf; i += 64) cnt[i] = 0;
    __syncthreads();
    const int q0 = pose_start[p], q1 = pose_start[p + 1];
    for (int q = q0 + lane; q < q1; q += 64) {
        const int k1 = pose_edges_l[q];
        if (E[k1].flags & 2) continue;
        const int l = E[k1].point;
        for (int q2 = pt_start[l]; q2 < pt_start[l + 1]; q2++) { const int h2 = sgx_ba_edge_row(E, hidx, pt_edges[q2]); if (h2 >= 0) atomicAdd(&cnt[h2], 1); }
    }
    __syncthreads();
    int run = 0, nb = 0;                                          /* jobs / non-empty blocks of the row before this chunk of 64 columns */
    const int jo = FILL ? job_off[h1] : 0, bo = FILL ? blk_off[h1] : 0;
    for (int base = 0; base < nf; base += 64) {
        const int i = base + lane, c = i < nf ? cnt[i] : 0;
        int s = c, b = c > 0 ? 1 : 0;
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(s, d, 64), u = __shfl_up(b, d, 64); if (lane >= d) { s += t; b += u; } }
        if (FILL && i < nf) { if (c > 0) blk_start[bo + nb + b - 1] = jo + run + s - c; cnt[i] = run + s - c; }
        run += __shfl(s, 63, 64); nb += __shfl(b, 63, 64);
    }
    if (!FILL) { if (lane == 0) { row_jobs[h1] = run; row_blks[h1] = nb; } return; }
    __syncthreads();
    SgxBaJob *out = jobs + jo;
    for (int qb = q0; qb < q1; qb += 64) {
        int my_k1 = -1, my_a0 = 0, my_a1 = 0;                     /* 64 edges of the row fetched side by side, then visited in order */
        if (qb + lane < q1) { my_k1 = pose_edges_l[qb + lane]; if (E[my_k1].flags & 2) my_k1 = -1; else { const int l = E[my_k1].point; my_a0 = pt_start[l]; my_a1 = pt_start[l + 1]; } }
        const int nq = min(64, q1 - qb);
        for (int j = 0; j < nq; j++) {
            const int k1 = __shfl(my_k1, j, 64), a0 = __shfl(my_a0, j, 64), a1 = __shfl(my_a1, j, 64);
            if (k1 < 0) continue;
            for (int t0 = a0; t0 < a1; t0 += 64) {
                const int q2 = t0 + lane, nt = min(64, a1 - t0);
                int k2 = -1, h2 = -1;
                if (q2 < a1) { k2 = pt_edges[q2]; h2 = sgx_ba_edge_row(E, hidx, k2); }
                int rank = 0, tot = 0;
                for (int i = 0; i < nt; i++) { const int v = __shfl(h2, i, 64); if (v == h2) { tot++; if (i < lane) rank++; } }
                int cur = 0;
                if (h2 >= 0) { cur = cnt[h2]; SgxBaJob jb; jb.k1 = k1; jb.k2 = k2; out[cur + rank] = jb; }
                __syncthreads();                                  /* one wave: orders the cursor reads above before the updates below */
                if (h2 >= 0 && rank == tot - 1) cnt[h2] = cur + tot;
                __syncthreads();
            }
        }
    }
}

// exclusive sums of the rows' job and block counts; tot = { jobs, blocks }; the end marker of the block list
__global__ void __launch_bounds__(256) k_ba_jobs_scan(int nf, const int *row_jobs, const int *row_blks, int *job_off, int *blk_off, int *tot, int *blk_start)
{
    __shared__ int sj[256], sb[256];
    const int tid = (int)threadIdx.x, per = (nf + 255) / 256, lo = min(nf, tid * per), hi = min(nf, lo + per);
    int aj = 0, ab = 0;
    for (int i = lo; i < hi; i++) { aj += row_jobs[i]; ab += row_blks[i]; }
    sj[tid] = aj; sb[tid] = ab;
    __syncthreads();
    if (tid == 0) { int rj = 0, rb = 0; for (int i = 0; i < 256; i++) { const int tj = sj[i], tb = sb[i]; sj[i] = rj; sb[i] = rb; rj += tj; rb += tb; } tot[0] = rj; tot[1] = rb; blk_start[rb] = rj; }
    __syncthreads();
    aj = sj[tid]; ab = sb[tid];
    for (int i = lo; i < hi; i++) { job_off[i] = aj; blk_off[i] = ab; aj += row_jobs[i]; ab += row_blks[i]; }
}
#endif

// k_ba_schur_init_env (round 6): the same values, written only where the envelope solver looks — the diagonal tiles and the tiles (r, k), r in R(k), of its column steps (the
// symbolic factorisation's pattern, which contains every coupled pose pair and every pose that straddles two tiles).  At 2 000 keyframes that is 16 MB instead of the 1.15 GB of the
// full 11 994 x 11 994 matrix, per LM trial.  Everything else in S stays undefined: k_ba_schur_pairs read-modify-writes the mirrored (upper) blocks there, nobody reads them
// (tests: the whole matrix poisoned with NaN first — sgx_ba_debug_set_init(2) — gives the same bits).  One workgroup per tile.
SGX_KERNEL(SGX_BA_THREADS) k_ba_schur_init_env(int nf, const double *Hpp, double lambda, double *S, double *coef, int nt, const int *rstart, const int *rows)
{
    SGX_THREADS_BEGIN(tid)
    const int NP = 6 * nf, t = (int)blockIdx.x;
    int tr = t, tc = t;
    if (t >= nt) {
        const int q = t - nt; int lo = 0, hi = nt;                // the column step of list entry q: the last k with rstart[k] <= q
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (rstart[mid] <= q) lo = mid; else hi = mid; }
        tc = lo; tr = rows[q];
    }
    for (int e = tid; e < SGX_NB * SGX_NB; e += SGX_BA_THREADS) {
        const int r = tr * SGX_NB + e / SGX_NB, c = tc * SGX_NB + e % SGX_NB;
        if (r < NP && c < NP) {
            double v = 0;
            if (r / 6 == c / 6) v = Hpp[(size_t)(r / 6) * 36 + 6 * (r % 6) + (c % 6)] + (r == c ? lambda : 0.0);
            S[(size_t)r * NP + c] = v;
        }
    }
    if (t == 0) for (int i = tid; i < NP; i += SGX_BA_THREADS) coef[i] = 0;
    SGX_THREADS_END
}

// k_ba_hpl_dinv (round 6): W_k = Hpl_k Dinv_l, once per active edge and trial (6 x 3 per edge) instead of once per job and destination column inside k_ba_schur_pairs (every job of
// edge k1 — about eight per edge, six columns each — recomputed the same three sums and fetched Dinv and the edge record for them).  The sums are the expressions k_ba_schur_pairs
// evaluated, in the same order, so the reduced system keeps its bits.
SGX_KERNEL(SGX_BA_THREADS) k_ba_hpl_dinv(long long n6, const SgxBaEdge *E, const int *hidx, const double *Hpl, const double *Dinv, double *W)
{
    SGX_THREADS_BEGIN(tid)
    const long long g = (long long)blockIdx.x * SGX_BA_THREADS + tid;
    if (g < n6) {
        const int k = (int)(g / 6), a = (int)(g % 6);
        const SgxBaEdge e = E[k];
        if (!(e.flags & 2) && hidx[e.pose] >= 0) {
            const double *Di = Dinv + (size_t)e.point * 9, *B1 = Hpl + (size_t)k * 18 + 3 * a;
            double *w = W + (size_t)k * 18 + 3 * a;
            w[0] = B1[0] * Di[0] + B1[1] * Di[3] + B1[2] * Di[6];
            w[1] = B1[0] * Di[1] + B1[1] * Di[4] + B1[2] * Di[7];
            w[2] = B1[0] * Di[2] + B1[1] * Di[5] + B1[2] * Di[8];
        }
    }
    SGX_THREADS_END
}

// Jobs arrive sorted by destination block (i1, i2) of the reduced system, landmark order inside a block (host: stable counting sort, once per active edge set).
// One thread per (destination block, entry): it starts from the value k_ba_schur_init left in S and subtracts the block's contributions ONE BY ONE IN LANDMARK
// ORDER — the order in which block_solver.hpp:380-433 visits them — and writes the entry once.  No atomics: the reduced system, hence the whole optimisation,
// is bit-reproducible from run to run.  The diagonal jobs (k1 == k2) of a diagonal block also accumulate coef(i1) += Hpl_1 (Dinv bl), same order.
SGX_KERNEL(SGX_BA_THREADS) k_ba_schur_pairs(long long nblk36, int nf, const int *blk_start, const SgxBaJob *jobs, const SgxBaEdge *E, const int *hidx,
                                            const double *bl, const double *Hpl, const double *W, double *S, double *coef)
{
    SGX_THREADS_BEGIN(tid)
    const long long g = (long long)blockIdx.x * SGX_BA_THREADS + tid;
    if (g < nblk36) {
        const int blk = (int)(g / 36), ent = (int)(g % 36), a = ent / 6, c = ent % 6;
        const int q0 = blk_start[blk], q1 = blk_start[blk + 1];
        const SgxBaJob j0 = jobs[q0];
        const int i1 = hidx[E[j0.k1].pose], i2 = hidx[E[j0.k2].pose], NP = 6 * nf;
        double acc = S[(size_t)(6 * i1 + a) * NP + 6 * i2 + c];
        double cacc = (i1 == i2 && c == 0) ? coef[6 * i1 + a] : 0.0;
        for (int q = q0; q < q1; q++) {
            const SgxBaJob jb = jobs[q];
            const double *W1 = W + (size_t)jb.k1 * 18 + 3 * a, *B2 = Hpl + (size_t)jb.k2 * 18 + 3 * c;      // W1 = row a of Hpl_1 Dinv (k_ba_hpl_dinv)
            const double bd0 = W1[0], bd1 = W1[1], bd2 = W1[2];
            acc += -(bd0 * B2[0] + bd1 * B2[1] + bd2 * B2[2]);
            if (jb.k1 == jb.k2 && c == 0) {
                const int l = E[jb.k1].point;
                const double b0 = bl[3 * (size_t)l], b1 = bl[3 * (size_t)l + 1], b2 = bl[3 * (size_t)l + 2];
                cacc += bd0 * b0 + bd1 * b1 + bd2 * b2;               // Hpl_1 Dinv bl
            }
        }
        S[(size_t)(6 * i1 + a) * NP + 6 * i2 + c] = acc;
        if (i1 == i2 && c == 0) coef[6 * i1 + a] = cacc;
    }
    SGX_THREADS_END
}

// k_ba_backsub: xl = Dinv (bl - Hpl^T xp) per landmark (block_solver.hpp:461-481)
SGX_KERNEL(SGX_BA_THREADS) k_ba_backsub(int nl, const int *pt_start, const int *pt_edges, const SgxBaEdge *E, const int *hidx, const uint8_t *pt_active,
                                        const double *bl, const double *Hpl, const double *Dinv, const double *xp, double *xl, const int *ok)
{
    if (!*ok) return;            // failed factorisation: keep the previous xl (g2o returns before the landmark update, block_solver.hpp:455-456)
    SGX_THREADS_BEGIN(tid)
    const int l = (int)blockIdx.x * SGX_BA_THREADS + tid;
    if (l < nl) {
        double c[3] = { bl[3 * (size_t)l], bl[3 * (size_t)l + 1], bl[3 * (size_t)l + 2] };
        for (int q = pt_start[l]; q < pt_start[l + 1]; q++) {
            const int k = pt_edges[q];
            const SgxBaEdge e = E[k];
            const int hp = hidx[e.pose];
            if ((e.flags & 2) || hp < 0) continue;
            const double *Bk = Hpl + (size_t)k * 18;
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                double s = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) s += Bk[3 * a + cc] * xp[6 * hp + a];
                c[cc] -= s;
            }
        }
        const double *Di = Dinv + (size_t)l * 9;
#pragma unroll
        for (int a = 0; a < 3; a++) xl[3 * (size_t)l + a] = pt_active[l] ? Di[3 * a] * c[0] + Di[3 * a + 1] * c[1] + Di[3 * a + 2] * c[2] : 0.0;
    }
    SGX_THREADS_END
}

// k_ba_update: push (backup) + oplus: T <- exp(xp) T for free poses (types_six_dof_expmap.h:73-76), X += xl (types_sba.h:52-56);
// also the per-block partial sums of computeScale (levenberg.cpp:182-189): sum x (lambda x + b)
SGX_KERNEL(SGX_BA_THREADS) k_ba_update(int np, int nl, const int *hidx, const uint8_t *pt_active, const double *xp, const double *xl, const double *bp, const double *bl,
                                       double lambda, SgxSE3 *T, double *X, SgxSE3 *Tb, double *Xb, double *partial)
{
    SGX_LDS double red[SGX_BA_THREADS];
    SGX_THREADS_BEGIN(tid)
    const int i = (int)blockIdx.x * SGX_BA_THREADS + tid;
    double sc = 0;
    if (i < np) {
        Tb[i] = T[i];
        const int hp = hidx[i];
        if (hp >= 0) {
            double u[6];
#pragma unroll
            for (int a = 0; a < 6; a++) { u[a] = xp[6 * hp + a]; sc += u[a] * (lambda * u[a] + bp[6 * hp + a]); }
            SgxSE3 ex, up; sgx_se3_exp(u, ex); sgx_se3_mul(ex, T[i], up); T[i] = up;
        }
    }
    if (i < nl) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double v = X[3 * (size_t)i + a];
            Xb[3 * (size_t)i + a] = v;
            if (pt_active[i]) { const double d = xl[3 * (size_t)i + a]; X[3 * (size_t)i + a] = v + d; sc += d * (lambda * d + bl[3 * (size_t)i + a]); }
        }
    }
    red[tid] = sc;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { double s = 0; for (int k = 0; k < SGX_BA_THREADS; k++) s += red[k]; partial[blockIdx.x] = s; }
    SGX_THREADS_END
}

// k_ba_classify: chi2 > th or depth <= 0 per edge (Optimizer.cc:672-702, :713-742) from the edge's CURRENT stored error
// (errors of excluded / rejected-trial edges are deliberately stale, as in the reference); mode 0: set level + drop Huber, mode 1: erase flags
SGX_KERNEL(SGX_BA_THREADS) k_ba_classify(int ne, SgxBaEdge *E, const SgxSE3 *T, const double *X, const double *err, int mode, uint8_t *erase)
{
    SGX_THREADS_BEGIN(tid)
    const int k = (int)blockIdx.x * SGX_BA_THREADS + tid;
    if (k < ne) {
        SgxBaEdge e = E[k];
        const int stereo = e.flags & 1;
        const double er[3] = { err[3 * (size_t)k], err[3 * (size_t)k + 1], stereo ? err[3 * (size_t)k + 2] : 0.0 };
        const double c2 = sgx_po_chi2(er, (double)e.info, stereo);
        const double Xl[3] = { X[3 * (size_t)e.point], X[3 * (size_t)e.point + 1], X[3 * (size_t)e.point + 2] };
        double p[3]; sgx_se3_map(T[e.pose], Xl, p);
        const bool bad = c2 > (stereo ? 7.815 : 5.991) || !(p[2] > 0.0);
        if (mode == 0) { if (bad) e.flags |= 2; e.flags &= ~4; E[k].flags = e.flags; }
        else erase[k] = bad ? 1 : 0;
    }
    SGX_THREADS_END
}

// float 4x4 <-> SE3Quat at the boundary (Converter.cc:37-71)
SGX_KERNEL(SGX_BA_THREADS) k_ba_poses_in(int np, const float *Tcw, SgxSE3 *T)
{
    SGX_THREADS_BEGIN(tid)
    const int i = (int)blockIdx.x * SGX_BA_THREADS + tid;
    if (i < np) { SgxSE3 s; sgx_se3_from_cv(Tcw + 16 * (size_t)i, s); T[i] = s; }
    SGX_THREADS_END
}
SGX_KERNEL(SGX_BA_THREADS) k_ba_poses_out(int np, const uint8_t *fixed, const SgxSE3 *T, float *Tcw)
{
    SGX_THREADS_BEGIN(tid)
    const int i = (int)blockIdx.x * SGX_BA_THREADS + tid;
    if (i < np && fixed[i] != 1) { float o[16]; sgx_se3_to_cv(T[i], o); for (int a = 0; a < 16; a++) Tcw[16 * (size_t)i + a] = o[a]; }
    SGX_THREADS_END
}

// k_ba_maxdiag: per-block max |diag| over the free poses' Hpp blocks and the active landmarks' Hll blocks
// (computeLambdaInit, levenberg.cpp:166-180)
SGX_KERNEL(SGX_BA_THREADS) k_ba_maxdiag(int nf, int nl, const double *Hpp, const double *Hll, const uint8_t *pt_active, double *partial)
{
    SGX_LDS double red[SGX_BA_THREADS];
    SGX_THREADS_BEGIN(tid)
    const int i = (int)blockIdx.x * SGX_BA_THREADS + tid;
    double m = 0;
    if (i < nf) { for (int j = 0; j < 6; j++) m = fmax(m, fabs(Hpp[(size_t)i * 36 + 7 * j])); }
    if (i < nl && pt_active[i]) { for (int j = 0; j < 3; j++) m = fmax(m, fabs(Hll[(size_t)i * 9 + 4 * j])); }
    red[tid] = m;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { double s = 0; for (int k = 0; k < SGX_BA_THREADS; k++) s = fmax(s, red[k]); partial[blockIdx.x] = s; }
    SGX_THREADS_END
}
