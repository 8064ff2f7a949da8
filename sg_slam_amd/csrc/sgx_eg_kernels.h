// sgx_eg_kernels.h — the optimisation of Optimizer::OptimizeEssentialGraph (src/sg-slam/src/Optimizer.cc:781-1042, :794 setUserLambdaInit(1e-16), :961-962 optimize(20)) on the device:
// EdgeSim3::computeError  _error = (C * v1 * v2^-1).log()  (G/types/types_seven_dof_expmap.h:100-108) with the NUMERIC Jacobians of BaseBinaryEdge for both vertices
// (G/core/base_binary_edge.hpp:131-205), information = identity, no robust kernel.  The normal equations are assembled densely (7 x 7 blocks) without atomics — diagonal blocks by a
// walk over each vertex's incident edges, off-diagonal blocks by a walk over the edges of each vertex pair, both in edge order — and solved by the blocked Cholesky of the bundle
// adjustment (chol_factor_solve); the Levenberg-Marquardt control runs on the host like LocalBundleAdjustment's.
#pragma once
#include "sgx_sim3.h"
#define SGX_EG_THREADS 256
#define SGX_EG_BLK 161                 /* doubles per edge: A_ii, A_ij, A_jj (49 each), b_i, b_j (7 each) */

SGX_DEV void sgx_eg_load(const double *p, SgxSim3 &s) { s.q[0] = p[0]; s.q[1] = p[1]; s.q[2] = p[2]; s.q[3] = p[3]; s.t[0] = p[4]; s.t[1] = p[5]; s.t[2] = p[6]; s.s = p[7]; }
SGX_DEV void sgx_eg_error(const SgxSim3 &C, const SgxSim3 &vi, const SgxSim3 &vj, double e[7])
{
    SgxSim3 a, inv, b; sgx_sim3_mul(C, vi, a); sgx_sim3_inverse(vj, inv); sgx_sim3_mul(a, inv, b); sgx_sim3_log(b, e);
}

// errors of all edges at V (kept for the linearisation) and the chi2 partial of every workgroup (summed on the host in block order)
SGX_KERNEL(SGX_EG_THREADS) k_eg_errors(int ne, const int *e_i, const int *e_j, const double *meas, const double *V, double *err, double *part_chi)
{
    SGX_LDS double red[SGX_EG_THREADS];
    SGX_THREADS_BEGIN(tid)
    const int k = (int)blockIdx.x * SGX_EG_THREADS + tid;
    double c = 0;
    if (k < ne) {
        SgxSim3 M, a, b; sgx_eg_load(meas + 8 * (size_t)k, M); sgx_eg_load(V + 8 * (size_t)e_i[k], a); sgx_eg_load(V + 8 * (size_t)e_j[k], b);
        double e[7]; sgx_eg_error(M, a, b, e);
        for (int r = 0; r < 7; r++) { err[7 * (size_t)k + r] = e[r]; c += e[r] * e[r]; }
    }
    red[tid] = c;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { double s = 0; for (int i = 0; i < SGX_EG_THREADS; i++) s += red[i]; part_chi[blockIdx.x] = s; }
    SGX_THREADS_END
}

// per edge: numeric Jacobians of both (free) vertices and the edge's contributions A_ii = Ji^T Ji, A_ij = Ji^T Jj, A_jj = Jj^T Jj, b_i = -Ji^T e, b_j = -Jj^T e
SGX_KERNEL(SGX_EG_THREADS) k_eg_linearize(int ne, const int *e_i, const int *e_j, const double *meas, const double *V, const int *hidx, int fix_scale, const double *err, double *blk)
{
    SGX_THREADS_BEGIN(tid)
    const int k = (int)blockIdx.x * SGX_EG_THREADS + tid;
    if (k < ne) {
        const int vi = e_i[k], vj = e_j[k];
        const bool fi = hidx[vi] >= 0, fj = hidx[vj] >= 0;
        SgxSim3 M, Si, Sj; sgx_eg_load(meas + 8 * (size_t)k, M); sgx_eg_load(V + 8 * (size_t)vi, Si); sgx_eg_load(V + 8 * (size_t)vj, Sj);
        double Ji[7][7], Jj[7][7];                                  // [row][column]
        const double dl = 1e-9, scalar = 1.0 / (2 * dl);
        for (int d = 0; d < 7; d++) {
            double add[7] = { 0, 0, 0, 0, 0, 0, 0 }, ep[7], em[7]; SgxSim3 pp, pm;
            for (int r = 0; r < 7; r++) { Ji[r][d] = 0; Jj[r][d] = 0; }
            if (fi) {
                add[d] = dl; sgx_sim3_oplus(Si, add, fix_scale, pp); add[d] = -dl; sgx_sim3_oplus(Si, add, fix_scale, pm);
                sgx_eg_error(M, pp, Sj, ep); sgx_eg_error(M, pm, Sj, em);
                for (int r = 0; r < 7; r++) Ji[r][d] = scalar * (ep[r] - em[r]);
            }
            if (fj) {
                add[d] = dl; sgx_sim3_oplus(Sj, add, fix_scale, pp); add[d] = -dl; sgx_sim3_oplus(Sj, add, fix_scale, pm);
                sgx_eg_error(M, Si, pp, ep); sgx_eg_error(M, Si, pm, em);
                for (int r = 0; r < 7; r++) Jj[r][d] = scalar * (ep[r] - em[r]);
            }
        }
        double *o = blk + (size_t)k * SGX_EG_BLK;
        const double *e = err + 7 * (size_t)k;
        for (int a = 0; a < 7; a++) {
            double si = 0, sj = 0;
            for (int r = 0; r < 7; r++) { si += Ji[r][a] * e[r]; sj += Jj[r][a] * e[r]; }
            o[147 + a] = -si; o[154 + a] = -sj;
            for (int c = 0; c < 7; c++) {
                double hii = 0, hij = 0, hjj = 0;
                for (int r = 0; r < 7; r++) { hii += Ji[r][a] * Ji[r][c]; hij += Ji[r][a] * Jj[r][c]; hjj += Jj[r][a] * Jj[r][c]; }
                o[7 * a + c] = hii; o[49 + 7 * a + c] = hij; o[98 + 7 * a + c] = hjj;
            }
        }
    }
    SGX_THREADS_END
}

// diagonal block and right-hand side of free vertex h: its incident edges (inc_edge, side 0 = the edge's vertex 0) in edge order.  One workgroup per vertex, thread = entry.
SGX_KERNEL(64) k_eg_assemble_diag(int NP, const int *inc_start, const int *inc_edge, const uint8_t *inc_side, const double *blk, double *H, double *b)
{
    SGX_THREADS_BEGIN(tid)
    const int h = (int)blockIdx.x;
    if (tid < 56) {
        double s = 0;
        for (int q = inc_start[h]; q < inc_start[h + 1]; q++) {
            const double *o = blk + (size_t)inc_edge[q] * SGX_EG_BLK;
            s += tid < 49 ? o[(inc_side[q] ? 98 : 0) + tid] : o[(inc_side[q] ? 154 : 147) + (tid - 49)];
        }
        if (tid < 49) H[(size_t)(7 * h + tid / 7) * NP + 7 * h + tid % 7] = s; else b[7 * h + (tid - 49)] = s;
    }
    SGX_THREADS_END
}

// off-diagonal blocks: the edges of one unordered vertex pair (lo < hi; pair_flip = the edge's vertex 0 is hi) in edge order -> H(lo, hi) and its transpose H(hi, lo)
SGX_KERNEL(64) k_eg_assemble_pairs(int NP, const int *pair_start, const int *pair_lo, const int *pair_hi, const int *pair_edge, const uint8_t *pair_flip, const double *blk, double *H)
{
    SGX_THREADS_BEGIN(tid)
    const int g = (int)blockIdx.x;
    if (tid < 49) {
        const int a = tid / 7, c = tid % 7;
        double s = 0;
        for (int q = pair_start[g]; q < pair_start[g + 1]; q++) {
            const double *o = blk + (size_t)pair_edge[q] * SGX_EG_BLK + 49;
            s += pair_flip[q] ? o[7 * c + a] : o[7 * a + c];          // A_ij of an edge whose vertex 0 is `hi` is the (hi, lo) block: transpose it
        }
        const int lo = pair_lo[g], hi = pair_hi[g];
        H[(size_t)(7 * lo + a) * NP + 7 * hi + c] = s; H[(size_t)(7 * hi + c) * NP + 7 * lo + a] = s;
    }
    SGX_THREADS_END
}

// S = H + lambda I (the factorisation works in place), rhs = b, coef = 0
SGX_KERNEL(SGX_EG_THREADS) k_eg_damp(int NP, const double *H, const double *b, double lambda, double *S, double *bp, double *coef)
{
    SGX_THREADS_BEGIN(tid)
    const size_t n2 = (size_t)NP * NP;
    for (size_t i = (size_t)blockIdx.x * SGX_EG_THREADS + tid; i < n2; i += (size_t)gridDim.x * SGX_EG_THREADS) {
        const size_t r = i / NP, c = i - r * NP;
        S[i] = H[i] + (r == c ? lambda : 0.0);
    }
    if (blockIdx.x == 0) for (int i = tid; i < NP; i += SGX_EG_THREADS) { bp[i] = b[i]; coef[i] = 0.0; }
    SGX_THREADS_END
}

// push + oplus of every free vertex with its slice of x; partial sums of x (lambda x + b) (computeScale, levenberg.cpp:193-201), summed on the host in block order
SGX_KERNEL(SGX_EG_THREADS) k_eg_update(int nv, const int *hidx, const double *x, const double *b, double lambda, int fix_scale, const int *ok, const double *Vb, double *V, double *part_scale)
{
    SGX_LDS double red[SGX_EG_THREADS];
    SGX_THREADS_BEGIN(tid)
    const int v = (int)blockIdx.x * SGX_EG_THREADS + tid;
    double sc = 0;
    if (v < nv) {
        const int h = hidx[v];
        SgxSim3 s; sgx_eg_load(Vb + 8 * (size_t)v, s);
        if (h >= 0) {
            double u[7];
            for (int a = 0; a < 7; a++) { u[a] = x[7 * h + a]; sc += u[a] * (lambda * u[a] + b[7 * h + a]); }
            SgxSim3 n; sgx_sim3_oplus(s, u, fix_scale, n); s = n;
        }
        double *o = V + 8 * (size_t)v;
        o[0] = s.q[0]; o[1] = s.q[1]; o[2] = s.q[2]; o[3] = s.q[3]; o[4] = s.t[0]; o[5] = s.t[1]; o[6] = s.t[2]; o[7] = s.s;
    }
    red[tid] = sc;
    SGX_THREADS_END
    SGX_SYNC();
    SGX_THREADS_BEGIN(tid)
    if (tid == 0) { double s = 0; for (int i = 0; i < SGX_EG_THREADS; i++) s += red[i]; part_scale[blockIdx.x] = s; }
    SGX_THREADS_END
    (void)ok;
}

// the map-point correction that follows the optimisation (Optimizer.cc:1004-1041): P' = correctedSwr.map(Srw.map(P)), r = the point's reference keyframe
SGX_KERNEL(SGX_EG_THREADS) k_eg_correct_points(int n, const float *xw, const int *ref, const double *Srw, const double *cSwr, float *out)
{
    SGX_THREADS_BEGIN(tid)
    const int i = (int)blockIdx.x * SGX_EG_THREADS + tid;
    if (i < n) {
        SgxSim3 a, c; sgx_eg_load(Srw + 8 * (size_t)ref[i], a); sgx_eg_load(cSwr + 8 * (size_t)ref[i], c);
        const double p[3] = { (double)xw[3 * (size_t)i], (double)xw[3 * (size_t)i + 1], (double)xw[3 * (size_t)i + 2] };
        double m[3], o[3]; sgx_sim3_map(a, p, m); sgx_sim3_map(c, m, o);
        out[3 * (size_t)i] = (float)o[0]; out[3 * (size_t)i + 1] = (float)o[1]; out[3 * (size_t)i + 2] = (float)o[2];
    }
    SGX_THREADS_END
}
