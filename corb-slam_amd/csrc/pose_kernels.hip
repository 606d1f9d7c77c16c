// pose_kernels.hip -- ONE workgroup runs a whole Optimizer::PoseOptimization (C/src/Optimizer.cc:272-485):
// all stages, all Levenberg-Marquardt iterations and trials, the 6x6 solve and the outlier classification, without a
// single host round trip.  Semantics follow oracle/orc_ba.c (g2o LM: G/core/optimization_algorithm_levenberg.cpp:61-189,
// BlockSolver with one 6x6 pose block, Huber kernel, stale edge chi2, float invz of the stereo edge).
//
//   edges       : one thread per edge slot (stride 256); points are fixed, so an edge carries its world point
//   reductions  : 21 (upper H) + 6 (b) + 1 (chi2) partial sums per lane, transposing butterfly (32 exchanges) + 4 waves
//   solve       : thread 0, LDL^T without pivoting like the oracle's dense solver, SE3 exp-map update
//   batch       : blockIdx.x = problem (frame); problems are independent (tracking threads of many clients)
#include "pose_internal.h"
#include "ba_math.h"
#include <cfloat>

#ifndef PO_T
#define PO_T 512          // one workgroup per frame; measured per call (400 / 1 750 observations): 256 threads 0.40 / 0.61 ms, 512: 0.40 / 0.55, 1024: 0.57 / 0.67
#endif
#define PO_W (PO_T / 64)

// lane l ends up with the wave total of value id(l) = bits (5,4,3,2,1) of l -> 16 b5 + 8 b4 + 4 b3 + 2 b2 + b1
__device__ __forceinline__ double wave_transpose_reduce32(double (&v)[32])
{
    const int lane = threadIdx.x & 63;
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8, h2 = lane & 4, h1 = lane & 2;
    double t16[16], t8[8], t4[4], t2[2];
#pragma unroll
    for (int j = 0; j < 16; j++) t16[j] = (h5 ? v[j + 16] : v[j]) + __shfl_xor(h5 ? v[j] : v[j + 16], 32);
#pragma unroll
    for (int j = 0; j < 8; j++) t8[j] = (h4 ? t16[j + 8] : t16[j]) + __shfl_xor(h4 ? t16[j] : t16[j + 8], 16);
#pragma unroll
    for (int j = 0; j < 4; j++) t4[j] = (h3 ? t8[j + 4] : t8[j]) + __shfl_xor(h3 ? t8[j] : t8[j + 4], 8);
#pragma unroll
    for (int j = 0; j < 2; j++) t2[j] = (h2 ? t4[j + 2] : t4[j]) + __shfl_xor(h2 ? t4[j] : t4[j + 2], 4);
    double tot = (h1 ? t2[1] : t2[0]) + __shfl_xor(h1 ? t2[0] : t2[1], 2);
    tot += __shfl_xor(tot, 1);
    return tot;
}

__device__ __forceinline__ double block_sum_po(double v, double* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < PO_W; w++) t += red[w];
    return t;
}

// camera-frame point, error and chi2 of one edge (EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose share the
// arithmetic of the binary edges, types_six_dof_expmap.cpp:141-160)
__device__ __forceinline__ double po_edge_error(const double* q, const double* t, const double* X, const double* z, double w, int dim,
                                                double fx, double fy, double cx, double cy, double bf, double* err, double* Xc)
{
    quat_rot(q, X, Xc);
    Xc[0] += t[0]; Xc[1] += t[1]; Xc[2] += t[2];
    if (dim == 2) {
        err[0] = z[0] - (Xc[0] / Xc[2] * fx + cx);
        err[1] = z[1] - (Xc[1] / Xc[2] * fy + cy);
        err[2] = 0;
        return w * (err[0] * err[0] + err[1] * err[1]);
    }
    const float invz = (float)(1.0 / Xc[2]);
    const double r0 = Xc[0] * invz * fx + cx;
    const double r1 = Xc[1] * invz * fy + cy;
    const double r2 = r0 - bf * invz;
    err[0] = z[0] - r0; err[1] = z[1] - r1; err[2] = z[2] - r2;
    return w * (err[0] * err[0] + err[1] * err[1] + err[2] * err[2]);
}

// dense LDL^T without pivoting of the symmetric 6x6 system (oracle ldlt_solve); solves S x = b in place
__device__ __forceinline__ int po_ldlt6(double* a, double* b)
{
    const int n = 6;
    for (int j = 0; j < n; j++) {
        double d = a[j * n + j];
        for (int k = 0; k < j; k++) d -= a[j * n + k] * a[j * n + k] * a[k * n + k];
        if (!(fabs(d) <= DBL_MAX) || d == 0.0) return 0;
        a[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = a[i * n + j];
            for (int k = 0; k < j; k++) s -= a[i * n + k] * a[j * n + k] * a[k * n + k];
            a[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= a[i * n + k] * b[k]; b[i] = s; }
    for (int i = 0; i < n; i++) b[i] /= a[i * n + i];
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= a[k * n + i] * b[k]; b[i] = s; }
    return 1;
}

// CACHED: every problem of the launch has at most PO_T * PO_EPT edges -- a thread keeps its edges (point, observation, weight, kind, active flag,
// last chi2) in registers for the whole call, so that the two sweeps of every LM iteration are not two rounds of global-memory latency
// (13 -> 10 us per iteration at 1 750 edges).  Otherwise the edges are re-read from global memory in every sweep.
#define PO_EPT 4
#define PO_FOR_EDGES(j, i) _Pragma("unroll") for (int j = 0; j < (CACHED ? PO_EPT : 1); j++) for (int i = tid + (CACHED ? j * PO_T : 0); i < nE; i += (CACHED ? nE : PO_T))
template <bool CACHED> __global__ __launch_bounds__(PO_T) void pose_opt_kernel(CorbPoseDev d)
{
    __shared__ double s_pose[7], s_pose0[7], s_bak[7];
    __shared__ double s_w[PO_W][32], s_tot[32], s_red[PO_W];
    __shared__ double s_lambda, s_ni, s_cur, s_ini, s_rho;
    __shared__ int s_ok2, s_again, s_ok, s_qmax, s_nbad, s_iters, s_trials, s_touched;
    const int prob = blockIdx.x, tid = threadIdx.x;
    const int e0 = d.edge_off[prob], nE = d.edge_off[prob + 1] - e0;
    const double fx = d.cam[5 * prob], fy = d.cam[5 * prob + 1], cx = d.cam[5 * prob + 2], cy = d.cam[5 * prob + 3], bf = d.cam[5 * prob + 4];
    const double* PT = d.pt + 3 * (size_t)e0; const double* OBS = d.obs + 3 * (size_t)e0; const double* W = d.w + e0;
    const unsigned char* DIM = d.dim + e0;
    double* LAST = d.last_chi2 + e0; unsigned char* ACT = d.active + e0;
    if (tid < 7) { s_pose[tid] = d.pose[7 * (size_t)prob + tid]; s_pose0[tid] = s_pose[tid]; }
    if (tid == 0) { s_iters = 0; s_trials = 0; s_touched = 0; }
    double cX[CACHED ? PO_EPT : 1][3], cZ[CACHED ? PO_EPT : 1][3], cW[CACHED ? PO_EPT : 1], cL[CACHED ? PO_EPT : 1];
    int cD[CACHED ? PO_EPT : 1], cA[CACHED ? PO_EPT : 1];
    PO_FOR_EDGES(j, i) {
        ACT[i] = 1;
        if (CACHED) {
#pragma unroll
            for (int k = 0; k < 3; k++) { cX[j][k] = PT[3 * i + k]; cZ[j][k] = OBS[3 * i + k]; }
            cW[j] = W[i]; cD[j] = DIM[i]; cA[j] = 1; cL[j] = 0.0;
        } else LAST[i] = 0.0;
    }
    __syncthreads();

    const int n_stages = d.stage_limit ? min(d.n_stages, d.stage_limit[prob]) : d.n_stages;
    for (int sg = 0; sg < n_stages; sg++) {
        const CorbBAStage& S = d.stages[sg];
        const int robust = S.robust;
        const double d2 = (double)S.huber_mono, d3 = (double)S.huber_stereo;
        if (S.reset_estimates) { __syncthreads(); if (tid < 7) s_pose[tid] = s_pose0[tid]; }
        {   // a pose with at least one active edge is "touched" (written back); otherwise it is passed through
            int any = 0;
            PO_FOR_EDGES(j, i) any |= CACHED ? cA[j] : (int)ACT[i];
            if (any) s_touched = 1;
        }
        if (tid == 0) { s_ok = 1; s_nbad = 0; s_lambda = -1.0; s_ni = 2.0; }
        __syncthreads();
        // ---------------- optimizer.optimize(S.iterations) ----------------
        for (int it = 0; it < S.iterations; it++) {
            if (!s_ok) break;
            // currentChi = activeRobustChi2() and buildSystem() in one sweep (same errors)
            double acc[32];
#pragma unroll
            for (int k = 0; k < 32; k++) acc[k] = 0.0;
            {
                double q[4], t[3];
#pragma unroll
                for (int k = 0; k < 4; k++) q[k] = s_pose[k];
#pragma unroll
                for (int k = 0; k < 3; k++) t[k] = s_pose[4 + k];
                PO_FOR_EDGES(j, i) {
                    if (!(CACHED ? cA[j] : (int)ACT[i])) continue;
                    const int D = CACHED ? cD[j] : (int)DIM[i];
                    const double wi = CACHED ? cW[j] : W[i];
                    double err[3], Xc[3], rho[2] = { 0.0, 1.0 };
                    const double chi = po_edge_error(q, t, CACHED ? cX[j] : PT + 3 * i, CACHED ? cZ[j] : OBS + 3 * i, wi, D, fx, fy, cx, cy, bf, err, Xc);
                    if (CACHED) cL[j] = chi; else LAST[i] = chi;
                    double wgt = wi;
                    if (robust) { huber(chi, D == 2 ? d2 : d3, rho); acc[27] += rho[0]; wgt *= rho[1]; }
                    else acc[27] += chi;
                    // d e / d pose (types_six_dof_expmap.cpp:118-131, 214-233)
                    const double x = Xc[0], y = Xc[1], z = Xc[2], z_2 = z * z;
                    double B[18];
                    B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
                    B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
                    if (D == 3) { B[12] = B[0] - bf * y / z_2; B[13] = B[1] + bf * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf / z_2; }
                    else { B[12] = B[13] = B[14] = B[15] = B[16] = B[17] = 0; err[2] = 0; }
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        acc[21 + a] += B[a] * (-wgt * err[0]) + B[6 + a] * (-wgt * err[1]) + B[12 + a] * (-wgt * err[2]);
#pragma unroll
                        for (int c = a; c < 6; c++, k++) acc[k] += B[a] * wgt * B[c] + B[6 + a] * wgt * B[6 + c] + B[12 + a] * wgt * B[12 + c];
                    }
                }
            }
            {
                const double tot = wave_transpose_reduce32(acc);
                const int lane = tid & 63;
                __syncthreads();
                if ((lane & 1) == 0) s_w[tid >> 6][lane >> 1] = tot;
                __syncthreads();
                if (tid < 32) { double t = 0; for (int w = 0; w < PO_W; w++) t += s_w[w][tid]; s_tot[tid] = t; }
                __syncthreads();
            }
            if (tid == 0) {
                s_cur = s_tot[27]; s_ini = s_tot[27];
                if (it == 0) {                                   // computeLambdaInit (:166-180)
                    double maxDiag = 0; int k = 0;
                    for (int a = 0; a < 6; a++) for (int c = a; c < 6; c++, k++) if (c == a) maxDiag = fmax(fabs(s_tot[k]), maxDiag);
                    s_lambda = 1e-5 * maxDiag; s_ni = 2.0; s_nbad = 0;
                }
                s_qmax = 0;
            }
            __syncthreads();
            // ---------------- trials ----------------
            do {
                if (tid == 0) {
                    for (int k = 0; k < 7; k++) s_bak[k] = s_pose[k];                               // push()
                    double A[36], xx[6]; int k = 0;
                    for (int a = 0; a < 6; a++) for (int c = a; c < 6; c++, k++) { A[a * 6 + c] = s_tot[k]; A[c * 6 + a] = s_tot[k]; }
                    for (int a = 0; a < 6; a++) { A[a * 6 + a] += s_lambda; xx[a] = s_tot[21 + a]; }
                    int ok2 = po_ldlt6(A, xx);
                    if (!ok2) for (int a = 0; a < 6; a++) xx[a] = 0.0;
                    double eq[4], et[3], q[4] = { s_pose[0], s_pose[1], s_pose[2], s_pose[3] }, t[3] = { s_pose[4], s_pose[5], s_pose[6] };
                    se3_exp(xx, eq, et);
                    se3_premul(eq, et, q, t);                                                       // oplus
                    for (int a = 0; a < 4; a++) s_pose[a] = q[a];
                    for (int a = 0; a < 3; a++) s_pose[4 + a] = t[a];
                    double scale = 0;
                    for (int a = 0; a < 6; a++) scale += xx[a] * (s_lambda * xx[a] + s_tot[21 + a]);   // computeScale (:182-189)
                    s_rho = scale + 1e-3;
                    s_ok2 = ok2;
                }
                __syncthreads();
                double part = 0;
                {
                    double q[4], t[3];
#pragma unroll
                    for (int k = 0; k < 4; k++) q[k] = s_pose[k];
#pragma unroll
                    for (int k = 0; k < 3; k++) t[k] = s_pose[4 + k];
                    PO_FOR_EDGES(j, i) {
                        if (!(CACHED ? cA[j] : (int)ACT[i])) continue;
                        const int D = CACHED ? cD[j] : (int)DIM[i];
                        double err[3], Xc[3], rho[2];
                        double c = po_edge_error(q, t, CACHED ? cX[j] : PT + 3 * i, CACHED ? cZ[j] : OBS + 3 * i, CACHED ? cW[j] : W[i], D, fx, fy, cx, cy, bf, err, Xc);
                        if (CACHED) cL[j] = c; else LAST[i] = c;
                        if (robust) { huber(c, D == 2 ? d2 : d3, rho); c = rho[0]; }
                        part += c;
                    }
                }
                const double sum = block_sum_po(part, s_red);
                if (tid == 0) {
                    double tempChi = s_ok2 ? sum : DBL_MAX;
                    double rho_lm = (s_cur - tempChi) / s_rho;
                    if (rho_lm > 0 && fabs(tempChi) <= DBL_MAX) {
                        double alpha = 1. - pow((2 * rho_lm - 1), 3);
                        alpha = fmin(alpha, 2. / 3.);
                        const double sf = fmax(1. / 3., alpha);
                        s_lambda *= sf; s_ni = 2; s_cur = tempChi;
                    } else {
                        s_lambda *= s_ni; s_ni *= 2;
                        for (int k = 0; k < 7; k++) s_pose[k] = s_bak[k];                           // pop()
                    }
                    s_qmax++; s_trials++;
                    s_rho = rho_lm;
                    s_again = (rho_lm < 0 && s_qmax < 10) ? 1 : 0;
                }
                __syncthreads();
            } while (s_again);
            if (tid == 0) {
                s_iters++;
                if (s_qmax == 10 || s_rho == 0) s_ok = 0;                                            // Terminate
                else {
                    if ((s_ini - s_cur) * 1e3 < s_ini) s_nbad++; else s_nbad = 0;                    // stop criterion (:155-161)
                    if (s_nbad >= 3) s_ok = 0;
                }
            }
            __syncthreads();
        }
        __syncthreads();
        // ---------------- classification after optimize() (Optimizer.cc:399-466, 768-797) ----------------
        {
            const bool need_eval = S.check_depth || S.recompute_inactive;
            double q[4], t[3];
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = s_pose[k];
#pragma unroll
            for (int k = 0; k < 3; k++) t[k] = s_pose[4 + k];
            PO_FOR_EDGES(j, i) {
                const int act = CACHED ? cA[j] : (int)ACT[i];
                const int D = CACHED ? cD[j] : (int)DIM[i];
                double fresh = 0, depth = 1;
                if (need_eval) { double err[3], Xc[3]; fresh = po_edge_error(q, t, CACHED ? cX[j] : PT + 3 * i, CACHED ? cZ[j] : OBS + 3 * i, CACHED ? cW[j] : W[i], D, fx, fy, cx, cy, bf, err, Xc); depth = Xc[2]; }
                if (!act && S.recompute_inactive) { if (CACHED) cL[j] = fresh; else LAST[i] = fresh; }
                if (!act && !S.allow_reactivate) continue;
                const double last = CACHED ? cL[j] : LAST[i];
                const double th = D == 2 ? (double)S.chi2_mono : (double)S.chi2_stereo;
                bool out = S.float_compare ? ((float)last > (float)th) : (last > th);
                if (S.check_depth && !(depth > 0.0)) out = true;
                ACT[i] = out ? 0 : 1;
                if (CACHED) cA[j] = out ? 0 : 1;
            }
        }
        __syncthreads();
    }
    int inl = 0;
    PO_FOR_EDGES(j, i) inl += CACHED ? cA[j] : (int)ACT[i];
    const double ninl = block_sum_po((double)inl, s_red);
    if (tid < 7) d.pose[7 * (size_t)prob + tid] = s_pose[tid];
    if (tid == 0) { int* c = d.counters + 4 * (size_t)prob; c[0] = s_iters; c[1] = s_trials; c[2] = s_touched; c[3] = (int)ninl; }
}

void pose_launch_optimize(const CorbPoseDev& d, int max_edges, hipStream_t s)
{
    if (max_edges <= PO_T * PO_EPT) hipLaunchKernelGGL(pose_opt_kernel<true>, dim3(d.n_problems), dim3(PO_T), 0, s, d);
    else hipLaunchKernelGGL(pose_opt_kernel<false>, dim3(d.n_problems), dim3(PO_T), 0, s, d);
}
