// corb_graph.cpp -- C-ABI host side of Optimizer::OptimizeEssentialGraph (see include/corb_accel.h): Levenberg control flow of
// g2o (G/core/optimization_algorithm_levenberg.cpp:61-189) with setUserLambdaInit(1e-16); device kernels in graph_kernels.hip;
// the dense (7 x free keyframes)^2 system is solved by the hand-written blocked Cholesky of dense_chol.hip.  No CPU compute fallback.
#include "graph_internal.h"
#include "corb_workspace.h"
#include "dense_chol.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cstring>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);


struct GPool : CorbScratch { GPool() : CorbScratch(1) {} };     // long-optimisation lane

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

extern "C" int corb_optimize_essential_graph(int n_keyframes, double* S, const uint8_t* fixed, int n_edges, const int32_t* vi, const int32_t* vj,
                                             const double* measurement, int iterations, int fix_scale, float* Tiw_out, int n_points,
                                             const int32_t* point_ref, float* points, double* chi2_hist, int32_t* iters_done, int device)
{
    const int K = n_keyframes, E = n_edges, M = n_points;
    if (K < 1 || !S || !fixed || E < 0 || (E > 0 && (!vi || !vj || !measurement)) || iterations < 0 || M < 0 || (M > 0 && (!point_ref || !points))) { corb_set_error("corb_optimize_essential_graph: bad argument"); return CORB_ERR_ARG; }
    for (int e = 0; e < E; e++) if (vi[e] < 0 || vi[e] >= K || vj[e] < 0 || vj[e] >= K) { corb_set_error("corb_optimize_essential_graph: edge %d out of range", e); return CORB_ERR_ARG; }
    std::vector<int> idx(K); int nP = 0;
    for (int k = 0; k < K; k++) idx[k] = fixed[k] ? -1 : nP++;
    const int sp = 7 * nP;
    if ((double)sp * sp * 16.0 > 200e9) { corb_set_error("corb_optimize_essential_graph: %d free keyframes need a %.0f GB dense system", nP, (double)sp * sp * 16e-9); return CORB_ERR_ARG; }
    int rc = corb_select_device(device); if (rc) return rc;
    GPool pool;
    if (!pool.stream) { corb_set_error("workspace: stream creation failed"); return CORB_ERR_HIP; }
    hipStream_t st = pool.stream;                       // every launch and copy of this optimisation runs on the workspace stream
    CorbGraphDev d; memset(&d, 0, sizeof(d));
    d.K = K; d.E = E; d.nP = nP; d.sp = sp; d.fix_scale = fix_scale ? 1 : 0;
    double *dV, *dV0, *dVbak, *dmeas, *dscal; unsigned char* dfixed; int *didx, *dvi, *dvj, *dinfo;
    HIPCHK(pool.upload(&dV, S, (size_t)8 * K)); HIPCHK(pool.upload(&dV0, S, (size_t)8 * K)); HIPCHK(pool.alloc(&dVbak, (size_t)8 * K));
    HIPCHK(pool.upload(&dfixed, fixed, (size_t)K)); HIPCHK(pool.upload(&didx, idx.data(), (size_t)K));
    HIPCHK(pool.upload(&dvi, vi, (size_t)E)); HIPCHK(pool.upload(&dvj, vj, (size_t)E)); HIPCHK(pool.upload(&dmeas, measurement, (size_t)8 * E));
    HIPCHK(pool.alloc(&d.H, (size_t)sp * sp)); HIPCHK(pool.alloc(&d.A, (size_t)sp * sp)); HIPCHK(pool.alloc(&d.b, (size_t)sp)); HIPCHK(pool.alloc(&d.x, (size_t)sp));
    const int nparts = std::max(1, std::min(256, (E + 255) / 256));
    HIPCHK(pool.alloc(&d.partial, (size_t)nparts)); HIPCHK(pool.alloc(&dscal, 4)); HIPCHK(pool.alloc(&dinfo, 1));
    double* chol_ws; HIPCHK(pool.alloc(&chol_ws, corb_chol_workspace_doubles(sp)));      // the dense solve's diagonal factors (dense_chol.h)
    d.V = dV; d.fixed = dfixed; d.idx = didx; d.vi = dvi; d.vj = dvj; d.meas = dmeas;
    // accumulation lists (the graph is fixed for the whole optimisation): incident edges per free vertex, edges per connected pair of free vertices
    {
        std::vector<int> voff(nP + 1, 0), vedge, plo, phi, poff, pedge;
        for (int e = 0; e < E; e++) {
            if (fixed[vi[e]] && fixed[vj[e]]) continue;                      // the edge kernel skips it (allVerticesFixed)
            if (idx[vi[e]] >= 0) voff[idx[vi[e]] + 1]++;
            if (idx[vj[e]] >= 0 && vj[e] != vi[e]) voff[idx[vj[e]] + 1]++;
        }
        for (int h = 0; h < nP; h++) voff[h + 1] += voff[h];
        vedge.resize(voff[nP]);
        { std::vector<int> cur(voff.begin(), voff.end() - 1);
          for (int e = 0; e < E; e++) {
              if (fixed[vi[e]] && fixed[vj[e]]) continue;
              if (idx[vi[e]] >= 0) vedge[cur[idx[vi[e]]]++] = (e << 1);
              if (idx[vj[e]] >= 0 && vj[e] != vi[e]) vedge[cur[idx[vj[e]]]++] = (e << 1) | 1;
          } }
        std::vector<std::pair<long long, int>> pk;                              // (lo * nP + hi, edge << 1 | flip), stable order = edge order inside a pair
        for (int e = 0; e < E; e++) {
            const int a = idx[vi[e]], c = idx[vj[e]];
            if (a < 0 || c < 0 || a == c) continue;
            const int lo = std::min(a, c), hi = std::max(a, c);
            pk.push_back({(long long)lo * nP + hi, (e << 1) | (a == lo ? 0 : 1)});
        }
        std::stable_sort(pk.begin(), pk.end(), [](const std::pair<long long, int>& x, const std::pair<long long, int>& y) { return x.first < y.first; });
        for (size_t t = 0; t < pk.size(); t++) {
            if (t == 0 || pk[t].first != pk[t - 1].first) { poff.push_back((int)t); plo.push_back((int)(pk[t].first / nP)); phi.push_back((int)(pk[t].first % nP)); }
            pedge.push_back(pk[t].second);
        }
        poff.push_back((int)pk.size());
        int *dvoff, *dvedge, *dplo, *dphi, *dpoff, *dpedge;
        HIPCHK(pool.upload(&dvoff, voff)); HIPCHK(pool.upload(&dvedge, vedge)); HIPCHK(pool.upload(&dplo, plo)); HIPCHK(pool.upload(&dphi, phi));
        HIPCHK(pool.upload(&dpoff, poff)); HIPCHK(pool.upload(&dpedge, pedge));
        HIPCHK(pool.alloc(&d.ejac, (size_t)105 * (E > 0 ? E : 1)));
        d.voff = dvoff; d.vedge = dvedge; d.n_pairs = (int)plo.size(); d.plo = dplo; d.phi = dphi; d.poff = dpoff; d.pedge = dpedge;
    }
    auto scalar = [&](int slot, double* out) -> int { HIPCHK(hipMemcpyAsync(out, dscal + slot, sizeof(double), hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); return CORB_OK; };
    auto chi2 = [&](double* out) -> int { eg_launch_chi2(d, nparts, dscal, st); return scalar(0, out); };
    double cur = 0;
    if (chi2_hist) { rc = chi2(&cur); if (rc) return rc; chi2_hist[0] = cur; }
    double lambda = 1e-16, ni = 2; int nBad = 0, it_done = 0; bool ok = true;
    for (int it = 0; it < iterations && ok && nP > 0; it++) {
        double currentChi; rc = chi2(&currentChi); if (rc) return rc;
        const double iniChi = currentChi; double tempChi = currentChi;
        eg_launch_build(d, st);
        if (it == 0) { lambda = 1e-16; ni = 2; nBad = 0; }                                   // setUserLambdaInit(1e-16) (Optimizer.cc:855)
        double rho = 0; int qmax = 0;
        do {
            HIPCHK(hipMemcpyAsync(dVbak, dV, sizeof(double) * 8 * (size_t)K, hipMemcpyDeviceToDevice, st));   // push()
            eg_launch_lambda(d, lambda, st);
            bool ok2 = true;
            // LinearSolverEigen on the dense (7 K)^2 system: hand-written blocked Cholesky + substitutions (dense_chol.hip)
            corb_launch_chol_solve(d.A, sp, sp, d.x, dinfo, chol_ws, st);
            HIPCHK(hipGetLastError());
            int info = 0; HIPCHK(hipMemcpyAsync(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
            ok2 = info == 0;                                                                   // not positive definite => solve() returns false
            if (!ok2) HIPCHK(hipMemsetAsync(d.x, 0, sizeof(double) * (size_t)sp, st));
            double scale = 0;
            eg_launch_update(d, lambda, dscal + 1, st);
            rc = scalar(1, &scale); if (rc) return rc;
            rc = chi2(&tempChi); if (rc) return rc;
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            scale += 1e-3; rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3); alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                HIPCHK(hipMemcpyAsync(dV, dVbak, sizeof(double) * 8 * (size_t)K, hipMemcpyDeviceToDevice, st));   // pop()
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        it_done++;
        if (chi2_hist) chi2_hist[it_done] = currentChi;
        if (qmax == 10 || rho == 0) { ok = false; continue; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) ok = false;
    }
    if (iters_done) *iters_done = it_done;
    // SE3 recovery and map point correction (Optimizer.cc:1045-1114) on the device, then one download
    float* dT = nullptr; float* dpts = nullptr; int* dref = nullptr;
    if (Tiw_out) HIPCHK(pool.alloc(&dT, (size_t)16 * K));
    if (M > 0) { HIPCHK(pool.upload(&dpts, points, (size_t)3 * M)); HIPCHK(pool.upload(&dref, point_ref, (size_t)M)); }
    if (Tiw_out || M > 0) {
        float* dTT = dT; if (!dTT) HIPCHK(pool.alloc(&dTT, (size_t)16 * K));
        eg_launch_apply(K, dV0, dV, dTT, M, dref, dpts, st);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        if (Tiw_out) HIPCHK(hipMemcpy(Tiw_out, dTT, sizeof(float) * 16 * (size_t)K, hipMemcpyDeviceToHost));
        if (M > 0) HIPCHK(hipMemcpy(points, dpts, sizeof(float) * 3 * (size_t)M, hipMemcpyDeviceToHost));
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(S, dV, sizeof(double) * 8 * (size_t)K, hipMemcpyDeviceToHost));
    return CORB_OK;
}
