// sim3_kernels.hip -- Optimizer::OptimizeSim3 (C/src/Optimizer.cc:1119-1311) as ONE workgroup per loop-closure candidate:
// both optimize() rounds, every Levenberg-Marquardt trial, the numeric Jacobians of g2o's EdgeSim3ProjectXYZ /
// EdgeInverseSim3ProjectXYZ (central differences, delta 1e-9: G/core/base_binary_edge.hpp:131-200), the 7x7 LDL^T and the
// chi2 re-classification run on the device without a host round trip.  Semantics follow oracle/orc_sim3.c
// (G/types/sim3.h exp-map / product / inverse, never re-normalised; VertexSim3Expmap::oplusImpl with _fix_scale).
#include "sim3_internal.h"
#include "lane_exchange.h"
#include "sim3_math.h"
#include <cfloat>

#define S3_T 256
#define S3_NV 36                      // 28 (upper 7x7) + 7 (b) + 1 (chi2)


// errors of the pair: e12 = obs1 - cam1(project(S * X2)), e21 = obs2 - cam2(project(S^-1 * X1))
__device__ __forceinline__ void s3_pair_errors(const double* S /* 8 */, const double* Si /* 8 */, const double* X1, const double* X2,
                                               const double* o1, const double* o2, const double* K, double* e12, double* e21)
{
    double r[3], m[3];
    quat_rot(S, X2, r);
    for (int i = 0; i < 3; i++) m[i] = S[7] * r[i] + S[4 + i];
    e12[0] = o1[0] - (m[0] / m[2] * K[0] + K[2]);
    e12[1] = o1[1] - (m[1] / m[2] * K[1] + K[3]);
    quat_rot(Si, X1, r);
    for (int i = 0; i < 3; i++) m[i] = Si[7] * r[i] + Si[4 + i];
    e21[0] = o2[0] - (m[0] / m[2] * K[4] + K[6]);
    e21[1] = o2[1] - (m[1] / m[2] * K[5] + K[7]);
}
__device__ __forceinline__ int s3_ldlt7(double* a, double* b)
{
    const int n = 7;
    for (int j = 0; j < n; j++) {
        double d = a[j * n + j];
        for (int k = 0; k < j; k++) d -= a[j * n + k] * a[j * n + k] * a[k * n + k];
        if (!(fabs(d) <= DBL_MAX) || d <= 0.0) return 0;                 // Eigen::LDLT::isPositive()
        a[j * n + j] = d;
        for (int i = j + 1; i < n; i++) { double s = a[i * n + j]; for (int k = 0; k < j; k++) s -= a[i * n + k] * a[j * n + k] * a[k * n + k]; a[i * n + j] = s / d; }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= a[i * n + k] * b[k]; b[i] = s; }
    for (int i = 0; i < n; i++) b[i] /= a[i * n + i];
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= a[k * n + i] * b[k]; b[i] = s; }
    return 1;
}

__global__ __launch_bounds__(S3_T) void sim3_opt_kernel(CorbSim3Dev d)
{
    __shared__ double s_S[8], s_bak[8];
    __shared__ double s_pert[15][2][8];            // [0] nominal, [1+d] +delta, [8+d] -delta ; [.][0] state, [.][1] inverse
    __shared__ double s_part[S3_NV][S3_T / 64];    // per-wave partial sums
    __shared__ double s_tot[S3_NV];
    __shared__ double s_lambda, s_ni, s_cur, s_ini, s_rho;
    __shared__ int s_ok2, s_again, s_ok, s_qmax, s_nbad, s_iters, s_trials;
    const int prob = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e0 = d.off[prob], n = d.off[prob + 1] - e0;
    double K[8];
    for (int i = 0; i < 8; i++) K[i] = (double)d.K[8 * prob + i];
    const double th2 = (double)d.th2, delta = (double)sqrtf(d.th2);          // const float deltaHuber = sqrt(th2)
    const int fix_scale = d.fix_scale;
    unsigned char* REM = d.removed + e0; double* L12 = d.last12 + e0; double* L21 = d.last21 + e0;
    if (tid < 8) s_S[tid] = d.S[8 * (size_t)prob + tid];
    if (tid == 0) { s_iters = 0; s_trials = 0; }
    for (int i = tid; i < n; i += S3_T) { REM[i] = 0; L12[i] = 0; L21[i] = 0; }
    __syncthreads();
    int nIn = 0, updated = 0;

    // block-wide sum of NV per-thread values into s_tot (fixed order: lanes by shuffles, then the 4 waves)
    auto reduce = [&](double (&v)[S3_NV], int nv) {
        for (int k = 0; k < nv; k++) {
            double x = v[k];
            x = lx_wave_sum(x);
            if (lane == 0) s_part[k][wave] = x;
        }
        __syncthreads();
        if (tid < nv) s_tot[tid] = s_part[tid][0] + s_part[tid][1] + s_part[tid][2] + s_part[tid][3];
        __syncthreads();
    };
    // activeRobustChi2 at the state in s_pert[0] (also records every edge's chi2)
    auto robust_chi2 = [&]() -> double {
        double part = 0;
        for (int i = tid; i < n; i += S3_T) {
            if (REM[i]) continue;
            const int g = e0 + i;
            const double X1[3] = { (double)d.p1c[3 * g], (double)d.p1c[3 * g + 1], (double)d.p1c[3 * g + 2] }, X2[3] = { (double)d.p2c[3 * g], (double)d.p2c[3 * g + 1], (double)d.p2c[3 * g + 2] };
            const double o1[2] = { (double)d.obs1[2 * g], (double)d.obs1[2 * g + 1] }, o2[2] = { (double)d.obs2[2 * g], (double)d.obs2[2 * g + 1] };
            double e12[2], e21[2], rho[2];
            s3_pair_errors(s_pert[0][0], s_pert[0][1], X1, X2, o1, o2, K, e12, e21);
            const double c12 = (double)d.w1[g] * (e12[0] * e12[0] + e12[1] * e12[1]), c21 = (double)d.w2[g] * (e21[0] * e21[0] + e21[1] * e21[1]);
            L12[i] = c12; L21[i] = c21;
            huber(c12, delta, rho); part += rho[0];
            huber(c21, delta, rho); part += rho[0];
        }
        double v[S3_NV]; v[0] = part;
        reduce(v, 1);
        return s_tot[0];
    };
    auto publish = [&](int slot, const S3State& S) {          // state + inverse into LDS
        S3State Si; s3_inv(S, Si);
        for (int k = 0; k < 4; k++) { s_pert[slot][0][k] = S.q[k]; s_pert[slot][1][k] = Si.q[k]; }
        for (int k = 0; k < 3; k++) { s_pert[slot][0][4 + k] = S.t[k]; s_pert[slot][1][4 + k] = Si.t[k]; }
        s_pert[slot][0][7] = S.s; s_pert[slot][1][7] = Si.s;
    };
    auto load_state = [&](S3State& S) { for (int k = 0; k < 4; k++) S.q[k] = s_S[k]; for (int k = 0; k < 3; k++) S.t[k] = s_S[4 + k]; S.s = s_S[7]; };

    for (int round = 0; round < 2; round++) {
        int iters = 5;
        if (round == 1) {
            // classification after the first optimize(): both edges of a pair go when either exceeds th2
            int bad = 0;
            for (int i = tid; i < n; i += S3_T) if (!REM[i] && (L12[i] > th2 || L21[i] > th2)) { REM[i] = 1; bad++; }
            double v[S3_NV]; v[0] = (double)bad; reduce(v, 1);
            const int nBad = (int)s_tot[0];
            iters = nBad > 0 ? 10 : 5;
            if (n - nBad < 10) break;                                      // return 0, g2oS12 untouched
        }
        if (tid == 0) { s_ok = 1; s_nbad = 0; s_lambda = -1.0; s_ni = 2.0; }
        __syncthreads();
        for (int it = 0; it < iters; it++) {
            if (!s_ok) break;
            // nominal + the 14 perturbed states of the numeric Jacobian (threads 0..14)
            if (tid < 15) {
                S3State S; load_state(S);
                if (tid > 0) { double u[7] = { 0, 0, 0, 0, 0, 0, 0 }; const int dd = (tid - 1) % 7; u[dd] = tid <= 7 ? 1e-9 : -1e-9; s3_oplus(S, u, fix_scale); }
                publish(tid, S);
            }
            __syncthreads();
            double acc[S3_NV];
#pragma unroll
            for (int k = 0; k < S3_NV; k++) acc[k] = 0.0;
            const double scalar = 1.0 / (2 * 1e-9);
            for (int i = tid; i < n; i += S3_T) {
                if (REM[i]) continue;
                const int g = e0 + i;
                const double X1[3] = { (double)d.p1c[3 * g], (double)d.p1c[3 * g + 1], (double)d.p1c[3 * g + 2] }, X2[3] = { (double)d.p2c[3 * g], (double)d.p2c[3 * g + 1], (double)d.p2c[3 * g + 2] };
                const double o1[2] = { (double)d.obs1[2 * g], (double)d.obs1[2 * g + 1] }, o2[2] = { (double)d.obs2[2 * g], (double)d.obs2[2 * g + 1] };
                double e12[2], e21[2], J12[14], J21[14];
                s3_pair_errors(s_pert[0][0], s_pert[0][1], X1, X2, o1, o2, K, e12, e21);
                for (int dd = 0; dd < 7; dd++) {
                    double a12[2], a21[2], b12[2], b21[2];
                    s3_pair_errors(s_pert[1 + dd][0], s_pert[1 + dd][1], X1, X2, o1, o2, K, a12, a21);
                    s3_pair_errors(s_pert[8 + dd][0], s_pert[8 + dd][1], X1, X2, o1, o2, K, b12, b21);
                    J12[dd] = scalar * (a12[0] - b12[0]); J12[7 + dd] = scalar * (a12[1] - b12[1]);
                    J21[dd] = scalar * (a21[0] - b21[0]); J21[7 + dd] = scalar * (a21[1] - b21[1]);
                }
                const double c12 = (double)d.w1[g] * (e12[0] * e12[0] + e12[1] * e12[1]), c21 = (double)d.w2[g] * (e21[0] * e21[0] + e21[1] * e21[1]);
                L12[i] = c12; L21[i] = c21;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const double* e = k ? e21 : e12; const double* J = k ? J21 : J12;
                    double w = k ? (double)d.w2[g] : (double)d.w1[g], rho[2];
                    huber(k ? c21 : c12, delta, rho);
                    acc[35] += rho[0];
                    w *= rho[1];
                    int idx = 0;
#pragma unroll
                    for (int a = 0; a < 7; a++) {
                        acc[28 + a] += J[a] * (-w * e[0]) + J[7 + a] * (-w * e[1]);
#pragma unroll
                        for (int c = a; c < 7; c++, idx++) acc[idx] += J[a] * w * J[c] + J[7 + a] * w * J[7 + c];
                    }
                }
            }
            reduce(acc, S3_NV);
            if (tid == 0) {
                s_cur = s_tot[35]; s_ini = s_tot[35];
                if (it == 0) {
                    double maxDiag = 0; int idx = 0;
                    for (int a = 0; a < 7; a++) for (int c = a; c < 7; c++, idx++) if (c == a) maxDiag = fmax(fabs(s_tot[idx]), maxDiag);
                    s_lambda = 1e-5 * maxDiag; s_ni = 2.0; s_nbad = 0;
                }
                s_qmax = 0;
            }
            __syncthreads();
            do {
                if (tid == 0) {
                    for (int k = 0; k < 8; k++) s_bak[k] = s_S[k];
                    double A[49], x[7]; int idx = 0;
                    for (int a = 0; a < 7; a++) for (int c = a; c < 7; c++, idx++) { A[a * 7 + c] = s_tot[idx]; A[c * 7 + a] = s_tot[idx]; }
                    for (int a = 0; a < 7; a++) { A[a * 7 + a] += s_lambda; x[a] = s_tot[28 + a]; }
                    const int ok2 = s3_ldlt7(A, x);
                    if (!ok2) for (int a = 0; a < 7; a++) x[a] = 0.0;
                    S3State S; load_state(S);
                    s3_oplus(S, x, fix_scale);
                    for (int k = 0; k < 4; k++) s_S[k] = S.q[k];
                    for (int k = 0; k < 3; k++) s_S[4 + k] = S.t[k];
                    s_S[7] = S.s;
                    publish(0, S);
                    double scale = 0;
                    for (int a = 0; a < 7; a++) scale += x[a] * (s_lambda * x[a] + s_tot[28 + a]);
                    s_rho = scale + 1e-3; s_ok2 = ok2;
                }
                __syncthreads();
                const double sum = robust_chi2();
                if (tid == 0) {
                    const double tempChi = s_ok2 ? sum : DBL_MAX;
                    const double rho_lm = (s_cur - tempChi) / s_rho;
                    if (rho_lm > 0 && fabs(tempChi) <= DBL_MAX) {
                        double alpha = 1. - pow((2 * rho_lm - 1), 3); alpha = fmin(alpha, 2. / 3.);
                        s_lambda *= fmax(1. / 3., alpha); s_ni = 2; s_cur = tempChi;
                    } else { s_lambda *= s_ni; s_ni *= 2; for (int k = 0; k < 8; k++) s_S[k] = s_bak[k]; }
                    s_qmax++; s_trials++;
                    s_rho = rho_lm;
                    s_again = (rho_lm < 0 && s_qmax < 10) ? 1 : 0;
                }
                __syncthreads();
            } while (s_again);
            if (tid == 0) {
                s_iters++;
                if (s_qmax == 10 || s_rho == 0) s_ok = 0;
                else { if ((s_ini - s_cur) * 1e3 < s_ini) s_nbad++; else s_nbad = 0; if (s_nbad >= 3) s_ok = 0; }
            }
            __syncthreads();
        }
        __syncthreads();
        if (round == 1) {
            int in = 0;
            for (int i = tid; i < n; i += S3_T) { if (REM[i]) continue; if (L12[i] > th2 || L21[i] > th2) REM[i] = 1; else in++; }
            double v[S3_NV]; v[0] = (double)in; reduce(v, 1);
            nIn = (int)s_tot[0]; updated = 1;
        }
    }
    if (updated && tid < 8) d.S[8 * (size_t)prob + tid] = s_S[tid];
    if (tid == 0) { int* c = d.counters + 4 * (size_t)prob; c[0] = s_iters; c[1] = s_trials; c[2] = nIn; c[3] = updated; }
}

void sim3_launch_optimize(const CorbSim3Dev& d, hipStream_t s)
{
    hipLaunchKernelGGL(sim3_opt_kernel, dim3(d.n_problems), dim3(S3_T), 0, s, d);
}
