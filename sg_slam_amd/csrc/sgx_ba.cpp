// sgx_ba.cpp — host side of the LocalBundleAdjustment C-ABI: flattening helpers (CSR by landmark / by pose), the
// Levenberg-Marquardt control loop (statement-for-statement OptimizationAlgorithmLevenberg::solve,
// G/core/optimization_algorithm_levenberg.cpp:61-164) and the kernel launches.  Reference: src/sg-slam/src/Optimizer.cc:453-778.
#include "sgx_ba_kernels.h"
#include "../../include/sgx.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define SGX_CHECK_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    fprintf(stderr, "sgx: HIP error %d (%s) at %s:%d\n", (int)_e, hipGetErrorString(_e), __FILE__, __LINE__); return SGX_ERR_DEVICE; } } while (0)

namespace {
struct Dev {
    std::vector<void *> all;
    ~Dev() { for (void *p : all) (void)hipFree(p); }
    template <class T> int alloc(T **p, size_t n) { void *q = nullptr; if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess) return SGX_ERR_NOMEM; all.push_back(q); *p = (T *)q; return SGX_OK; }
};

struct BA {
    int np, nl, ne, nf, NP;
    SgxCam cam; double dMono, dStereo;
    const volatile int32_t *stop;
    // device
    SgxBaEdge *E; SgxSE3 *T, *Tb; double *X, *Xb, *err, *Hll, *bl, *Hpl, *Hpp, *bp, *S, *coef, *xp, *xl, *Dinv, *dwork, *partial;
    int *pt_start, *pt_edges, *pose_start, *pose_edges, *hidx, *free_pose, *ok; uint8_t *pt_active;
    int nblk_e, nblk_v;
    std::vector<double> hpart;
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
    SGX_LAUNCH(k_ba_errors, dim3(B.nblk_e), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.ne, B.E, B.T, B.X, B.cam, B.dMono, B.dStereo, B.err, B.partial);
    return sum_partials(B, B.nblk_e, chi);
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
        SGX_LAUNCH(k_ba_linearize_points, dim3((B.nl + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nl, B.pt_start, B.pt_edges,
                   B.E, B.T, B.X, B.hidx, B.err, B.cam, B.dMono, B.dStereo, B.Hll, B.bl, B.Hpl, B.pt_active);
        if (B.nf > 0)
            SGX_LAUNCH(k_ba_linearize_poses, dim3(B.nf), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.np, B.free_pose, B.pose_start, B.pose_edges, B.E, B.T, B.X, B.err,
                       B.cam, B.dMono, B.dStereo, B.Hpp, B.bp);
        if (it == 0) {
            SGX_LAUNCH(k_ba_maxdiag, dim3(B.nblk_v), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nf, B.nl, B.Hpp, B.Hll, B.pt_active, B.partial);
            double maxd = 0; rc = sum_partials(B, B.nblk_v, &maxd, true); if (rc != SGX_OK) return rc;
            lambda = 1e-5 * maxd; ni = 2; nBadLM = 0;
        }
        double rho = 0; int qmax = 0;
        do {
            int ok2 = 1;
            if (B.NP > 0) {
                const int g = (int)(((size_t)B.NP * B.NP + SGX_BA_THREADS - 1) / SGX_BA_THREADS);
                SGX_LAUNCH(k_ba_schur_init, dim3(g > 4096 ? 4096 : g), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nf, B.Hpp, lambda, B.S, B.coef);
            }
            SGX_LAUNCH(k_ba_schur, dim3((B.nl + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nl, B.nf, B.pt_start, B.pt_edges, B.E,
                       B.hidx, B.pt_active, B.Hll, B.bl, B.Hpl, lambda, B.Dinv, B.S, B.coef);
            if (B.NP > 0) {                                          // blocked Cholesky of the reduced camera system
                const int one = 1;
                SGX_CHECK_HIP(hipMemcpy(B.ok, &one, 4, hipMemcpyHostToDevice));
                const int nt = (B.NP + SGX_NB - 1) / SGX_NB;
                for (int kb = 0; kb < nt; kb++) {
                    const int k0 = kb * SGX_NB, rem = nt - kb - 1;
                    SGX_LAUNCH(k_chol_diag, dim3(1), dim3(SGX_NB * SGX_NB / 4), (sgx_stream_t)0, B.NP, k0, B.S, B.ok);
                    if (rem > 0) {
                        SGX_LAUNCH(k_chol_panel, dim3(rem), dim3(256), (sgx_stream_t)0, B.NP, k0, B.S, B.ok);
                        SGX_LAUNCH(k_chol_update, dim3(rem * (rem + 1) / 2), dim3(256), (sgx_stream_t)0, B.NP, k0, B.S, B.ok);
                    }
                }
                SGX_LAUNCH(k_chol_solve, dim3(1), dim3(1024), (sgx_stream_t)0, B.NP, B.S, B.bp, B.coef, B.xp, B.ok);
                SGX_CHECK_HIP(hipMemcpy(&ok2, B.ok, 4, hipMemcpyDeviceToHost));
            }
            if (ok2)
                SGX_LAUNCH(k_ba_backsub, dim3((B.nl + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.nl, B.pt_start, B.pt_edges, B.E, B.hidx,
                           B.pt_active, B.bl, B.Hpl, B.Dinv, B.xp, B.xl);
            // push + update + computeScale
            SGX_LAUNCH(k_ba_update, dim3(B.nblk_v), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.np, B.nl, B.hidx, B.pt_active, B.xp, B.xl, B.bp, B.bl, lambda,
                       B.T, B.X, B.Tb, B.Xb, B.partial);
            double scale = 0; rc = sum_partials(B, B.nblk_v, &scale); if (rc != SGX_OK) return rc;
            rc = active_chi2(B, &tempChi); if (rc != SGX_OK) return rc;
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

extern "C" int sgx_local_bundle_adjustment(const sgx_ba_problem *P, const sgx_camera *cam, const volatile int32_t *stop_flag,
                                           uint8_t *edge_erase, sgx_ba_stats *stats)
{
    if (!P || !cam || !edge_erase || P->n_poses < 1 || P->n_points < 1 || P->n_edges < 1 || !P->poses || !P->pose_fixed || !P->points ||
        !P->edge_pose || !P->edge_point || !P->edge_obs || !P->edge_info) return SGX_ERR_INVALID;
    BA B; memset((void *)&B, 0, offsetof(BA, hpart));
    B.np = P->n_poses; B.nl = P->n_points; B.ne = P->n_edges; B.stop = stop_flag;
    B.cam.fx = cam->fx; B.cam.fy = cam->fy; B.cam.cx = cam->cx; B.cam.cy = cam->cy; B.cam.bf = cam->bf;
    B.dMono = (double)(float)sqrt(5.991); B.dStereo = (double)(float)sqrt(7.815);          // Optimizer.cc:569-570 (float)
    if (stats) memset(stats, 0, sizeof *stats);
    memset(edge_erase, 0, B.ne);
    if (stop_flag && *stop_flag) return SGX_OK;                                              // Optimizer.cc:655-657
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
    std::vector<double> Xd(3 * (size_t)B.nl);
    for (size_t i = 0; i < Xd.size(); i++) Xd[i] = (double)P->points[i];
    // ---- device state
    Dev D; float *dTcw = nullptr; uint8_t *dfixed = nullptr, *derase = nullptr;
    const int nv = B.np > B.nl ? B.np : B.nl;
    B.nblk_e = (B.ne + SGX_BA_THREADS - 1) / SGX_BA_THREADS; B.nblk_v = (nv + SGX_BA_THREADS - 1) / SGX_BA_THREADS;
    const int npart = B.nblk_e > B.nblk_v ? B.nblk_e : B.nblk_v;
    int rc = SGX_OK;
#define A(p, n) if ((rc = D.alloc(&(p), (n))) != SGX_OK) return rc
    A(B.E, B.ne); A(B.T, B.np); A(B.Tb, B.np); A(B.X, 3 * (size_t)B.nl); A(B.Xb, 3 * (size_t)B.nl); A(B.err, 3 * (size_t)B.ne);
    A(B.Hll, 9 * (size_t)B.nl); A(B.bl, 3 * (size_t)B.nl); A(B.Hpl, 18 * (size_t)B.ne); A(B.Hpp, 36 * (size_t)B.nf); A(B.bp, B.NP);
    A(B.S, (size_t)B.NP * B.NP); A(B.coef, B.NP); A(B.xp, B.NP); A(B.xl, 3 * (size_t)B.nl); A(B.Dinv, 9 * (size_t)B.nl); A(B.dwork, B.NP);
    A(B.partial, npart); A(B.pt_start, B.nl + 1); A(B.pt_edges, B.ne); A(B.pose_start, B.np + 1); A(B.pose_edges, B.ne); A(B.hidx, B.np);
    A(B.free_pose, B.nf); A(B.ok, 1); A(B.pt_active, B.nl); A(dTcw, 16 * (size_t)B.np); A(dfixed, B.np); A(derase, B.ne);
#undef A
    SGX_CHECK_HIP(hipMemcpy(B.E, E.data(), sizeof(SgxBaEdge) * B.ne, hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(B.X, Xd.data(), sizeof(double) * Xd.size(), hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(B.pt_start, pt_start.data(), 4 * (size_t)(B.nl + 1), hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(B.pt_edges, pt_edges.data(), 4 * (size_t)B.ne, hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(B.pose_start, pose_start.data(), 4 * (size_t)(B.np + 1), hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(B.pose_edges, pose_edges.data(), 4 * (size_t)B.ne, hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(B.hidx, hidx.data(), 4 * (size_t)B.np, hipMemcpyHostToDevice));
    if (B.nf) SGX_CHECK_HIP(hipMemcpy(B.free_pose, free_pose.data(), 4 * (size_t)B.nf, hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(dTcw, P->poses, 64 * (size_t)B.np, hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemcpy(dfixed, P->pose_fixed, B.np, hipMemcpyHostToDevice));
    SGX_CHECK_HIP(hipMemset(B.xp, 0, sizeof(double) * (B.NP ? B.NP : 1)));
    SGX_CHECK_HIP(hipMemset(B.xl, 0, sizeof(double) * 3 * (size_t)B.nl));
    SGX_CHECK_HIP(hipMemset(B.err, 0, sizeof(double) * 3 * (size_t)B.ne));
    SGX_CHECK_HIP(hipMemset(B.Hpl, 0, sizeof(double) * 18 * (size_t)B.ne));
    SGX_LAUNCH(k_ba_poses_in, dim3((B.np + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.np, dTcw, B.T);

    int it1 = 0, it2 = 0; double chi1 = 0, chi2 = 0;
    rc = optimize(B, 5, &it1, &chi1); if (rc != SGX_OK) return rc;                          // Optimizer.cc:659-660
    if (!stopped(B)) {                                                                      // :662-707
        SGX_LAUNCH(k_ba_classify, dim3(B.nblk_e), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.ne, B.E, B.T, B.X, B.err, 0, derase);
        rc = optimize(B, 10, &it2, &chi2); if (rc != SGX_OK) return rc;
    }
    SGX_LAUNCH(k_ba_classify, dim3(B.nblk_e), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.ne, B.E, B.T, B.X, B.err, 1, derase);   // :709-742
    SGX_LAUNCH(k_ba_poses_out, dim3((B.np + SGX_BA_THREADS - 1) / SGX_BA_THREADS), dim3(SGX_BA_THREADS), (sgx_stream_t)0, B.np, dfixed, B.T, dTcw);
    SGX_CHECK_HIP(hipGetLastError());
    SGX_CHECK_HIP(hipMemcpy(edge_erase, derase, B.ne, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(P->poses, dTcw, 64 * (size_t)B.np, hipMemcpyDeviceToHost));
    SGX_CHECK_HIP(hipMemcpy(Xd.data(), B.X, sizeof(double) * Xd.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < Xd.size(); i++) P->points[i] = (float)Xd[i];                      // Converter::toCvMat(Vector3d), Optimizer.cc:771-777
    if (stats) { stats->iterations_first = it1; stats->iterations_second = it2; stats->chi2_first = chi1; stats->chi2_second = chi2; stats->free_poses = B.nf; }
    return SGX_OK;
}
