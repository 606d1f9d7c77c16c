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
#include "lane_exchange.h"
#include <cfloat>

#ifndef PO_T
#define PO_T 512          // one workgroup per frame; measured per call (400 / 1 750 observations): 256 threads 0.40 / 0.61 ms, 512: 0.40 / 0.55, 1024: 0.57 / 0.67
#endif
#define PO_W (PO_T / 64)

// (cross-lane moves: lane_exchange.h -- round 6: an instantiation had 88 ds_bpermute_b32, 64 of them in the butterfly below that runs once per LM iteration; its halving
// stages now run on DPP moves and 4 remain there: 0.244 -> 0.226 ms per call at 400 observations, 0.285 -> 0.267 at 1 750)
// lane l ends up with the wave total of value id(l) = 16 b0 + 8 b1 + 4 b2 + 2 b3 + b4 (b_k = bit k of l; both halves of the wave hold every total).  The halving stages run
// on the masks 1, 2, 4, 8 -- DPP moves -- and only the last two exchanges (ONE value each: xor 16, xor 32) cross the rows of 16 lanes on the LDS crossbar; the first form
// of this butterfly halved on 32, 16 first: 48 of its 64 ds_bpermute_b32 sat there.  (The pairing order of a sum changed with it -- lane l with l ^ 1 first instead of
// l ^ 32 first --, i.e. the last bits of H and b: the estimates stay within the 1e-4 bar against the oracle, tests/test_gpu_staged.py; the routes that share this kernel stay
// bit-equal to each other.)
__device__ __forceinline__ double wave_transpose_reduce32(double (&v)[32])
{
    const int lane = threadIdx.x & 63;
    const bool h0 = lane & 1, h1 = lane & 2, h2 = lane & 4, h3 = lane & 8;
    double t16[16], t8[8], t4[4], t2[2];
#pragma unroll
    for (int j = 0; j < 16; j++) t16[j] = (h0 ? v[j + 16] : v[j]) + lx_xor<1>(h0 ? v[j] : v[j + 16]);
#pragma unroll
    for (int j = 0; j < 8; j++) t8[j] = (h1 ? t16[j + 8] : t16[j]) + lx_xor<2>(h1 ? t16[j] : t16[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; j++) t4[j] = (h2 ? t8[j + 4] : t8[j]) + lx_xor<4>(h2 ? t8[j] : t8[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; j++) t2[j] = (h3 ? t4[j + 2] : t4[j]) + lx_xor<8>(h3 ? t4[j] : t4[j + 2]);
    double tot = lx_xadd16(t2[0], t2[1]);
    tot = lx_xadd32(tot, tot);
    return tot;
}
__device__ __forceinline__ double block_sum_po(double v, double* red)
{
    v = lx_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < PO_W; w++) t += red[w];
    return t;
}

// camera-frame point, error and chi2 of one edge (EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose share the
// arithmetic of the binary edges, types_six_dof_expmap.cpp:141-160).  The whole call runs on ONE compute unit and is bound by its FP64 issue
// rate (4 cycles per wave64 instruction), so the per-edge instruction count is what matters: the rotation is applied as a matrix built once
// per sweep (9 fused multiply-adds instead of the quaternion form's 27 separate operations) and the sums below use explicit fma().
struct PoRt { double R[9], t[3]; };
__device__ __forceinline__ PoRt po_rt(const double* pose)
{
    PoRt o;
    const double x = pose[0], y = pose[1], z = pose[2], w = pose[3];
    o.R[0] = 1 - 2 * (y * y + z * z); o.R[1] = 2 * (x * y - z * w); o.R[2] = 2 * (x * z + y * w);
    o.R[3] = 2 * (x * y + z * w); o.R[4] = 1 - 2 * (x * x + z * z); o.R[5] = 2 * (y * z - x * w);
    o.R[6] = 2 * (x * z - y * w); o.R[7] = 2 * (y * z + x * w); o.R[8] = 1 - 2 * (x * x + y * y);
    o.t[0] = pose[4]; o.t[1] = pose[5]; o.t[2] = pose[6];
    // the pose is the same on every lane: in scalar registers (24 of them) it leaves the vector registers to the sweep's 28 sums and the cached edges
    // (the kernel runs at its 256-register budget: round 5 found 19 spilled registers inside the sweeps)
#pragma unroll
    for (int k = 0; k < 9; k++) o.R[k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(o.R[k])), __builtin_amdgcn_readfirstlane(__double2loint(o.R[k])));
#pragma unroll
    for (int k = 0; k < 3; k++) o.t[k] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(o.t[k])), __builtin_amdgcn_readfirstlane(__double2loint(o.t[k])));
    return o;
}
__device__ __forceinline__ double po_edge_error(const PoRt& P, const double* X, const double* z, double w, int dim,
                                                double fx, double fy, double cx, double cy, double bf, double* err, double* Xc)
{
#pragma unroll
    for (int k = 0; k < 3; k++) Xc[k] = fma(P.R[3 * k], X[0], fma(P.R[3 * k + 1], X[1], fma(P.R[3 * k + 2], X[2], P.t[k])));
    if (dim == 2) {
        const double iz = 1.0 / Xc[2];
        err[0] = z[0] - fma(Xc[0] * iz, fx, cx);
        err[1] = z[1] - fma(Xc[1] * iz, fy, cy);
        err[2] = 0;
        return w * fma(err[0], err[0], err[1] * err[1]);
    }
    const float invz = (float)(1.0 / Xc[2]);
    const double r0 = fma(Xc[0] * invz, fx, cx);
    const double r1 = fma(Xc[1] * invz, fy, cy);
    const double r2 = r0 - bf * invz;
    err[0] = z[0] - r0; err[1] = z[1] - r1; err[2] = z[2] - r2;
    return w * fma(err[0], err[0], fma(err[1], err[1], err[2] * err[2]));
}

// acc[0..20] (upper triangle of H, row-major) += J' w J and acc[21..26] += J' e for ONE row of the 3x6 Jacobian given by its 5 non-zero entries:
// ROW1 = false: pose coordinates 0 1 2 3 5 (rows 0 and 2, column 4 is zero); ROW1 = true: 0 1 2 4 5 (row 1, column 3 is zero)
template <int ROW1> __device__ __forceinline__ void po_row(const double (&J)[5], double wgt, double e, double (&acc)[32])
{
    constexpr int C[2][6] = { { 0, 1, 2, 3, -1, 4 }, { 0, 1, 2, -1, 3, 4 } };
    double w[5];
#pragma unroll
    for (int a = 0; a < 5; a++) w[a] = wgt * J[a];
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        if (C[ROW1][a] >= 0) acc[21 + a] = fma(J[C[ROW1][a]], e, acc[21 + a]);
#pragma unroll
        for (int c = a; c < 6; c++, k++)
            if (C[ROW1][a] >= 0 && C[ROW1][c] >= 0) acc[k] = fma(w[C[ROW1][a]], J[C[ROW1][c]], acc[k]);
    }
}

// 1 / x and 1 / sqrt(x) to double precision from the hardware's approximations and two Newton steps each (5 and 8 dependent instructions; an IEEE division is
// ~15 with its scaling and fix-up, `1.0 / sqrt(x)` twice that).  The section below runs on ONE lane between two sweeps of the whole workgroup -- cycle stamps
// (round 5): 4 800 cycles per LM trial, 40 % of a 400-observation call's 490 k and 28 % of a 1 750-observation call's 555 k -- so its instruction count is what counts.
__device__ __forceinline__ double po_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double po_rsqrt(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
}
// dense LDL^T without pivoting of the symmetric 6x6 system (oracle ldlt_solve); solves S x = b in place: one reciprocal per pivot.
__device__ __forceinline__ int po_ldlt6(double* a, double* b)
{
    const int n = 6;
    double inv[6];
    // (round 6) ld[j][k] = L_jk d_k is kept beside L: a term of the sums below is ONE fused multiply-add (the three separate operations of `s -= L_ik L_jk d_k`
    // were a third of this one-lane section's ~500 dependent FP64 instructions); the sums run in the same order k = 0, 1, ...
    double ld[6][6];
#pragma unroll
    for (int j = 0; j < n; j++) {
        double d = a[j * n + j];
#pragma unroll
        for (int k = 0; k < j; k++) d = fma(-a[j * n + k], ld[j][k], d);
        if (!(fabs(d) <= DBL_MAX) || d == 0.0) return 0;
        a[j * n + j] = d; inv[j] = po_rcp(d);
#pragma unroll
        for (int i = j + 1; i < n; i++) {
            double s = a[i * n + j];
#pragma unroll
            for (int k = 0; k < j; k++) s = fma(-a[i * n + k], ld[j][k], s);
            ld[i][j] = s;                                       // L_ij d_j
            a[i * n + j] = s * inv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < n; i++) { double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s = fma(-a[i * n + k], b[k], s); b[i] = s; }
#pragma unroll
    for (int i = 0; i < n; i++) b[i] *= inv[i];
#pragma unroll
    for (int i = n - 1; i >= 0; i--) { double s = b[i];
#pragma unroll
        for (int k = i + 1; k < n; k++) s = fma(-a[k * n + i], b[k], s); b[i] = s; }
    return 1;
}

// (q, t) <- exp(u) * (q, t) (SE3Quat::exp + operator*, se3quat.h:102-108, 223-257) on the same critical path.  With s = theta^2 the three coefficients of the
// exponential map -- a = sin(theta) / theta, b = (1 - cos theta) / theta^2, c = (theta - sin theta) / theta^3 -- are power series in s (six terms: the first one left
// out is below 1e-18 for theta < 0.1; no cancellation at small angles, no sincos, no square root of s, no division by theta); an LM step of a tracked frame is far below that.
// The rotation part of exp(u) goes straight to the quaternion (sin(theta/2) / theta * omega, cos(theta/2)): cos(theta/2) = sqrt((1 + cos theta) / 2),
// sin(theta/2) / theta = a / (2 cos(theta/2)).  Larger steps take ba_math.h's form.
__device__ __forceinline__ void po_oplus(const double* u, double* q, double* t)
{
    const double s = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    if (!(s < 0.01)) { double eq[4], et[3]; se3_exp(u, eq, et); se3_premul(eq, et, q, t); return; }
    const double a = fma(s, fma(s, fma(s, fma(s, fma(s, -1.0 / 39916800.0, 1.0 / 362880.0), -1.0 / 5040.0), 1.0 / 120.0), -1.0 / 6.0), 1.0);
    const double b = fma(s, fma(s, fma(s, fma(s, fma(s, -1.0 / 479001600.0, 1.0 / 3628800.0), -1.0 / 40320.0), 1.0 / 720.0), -1.0 / 24.0), 0.5);
    const double c = fma(s, fma(s, fma(s, fma(s, fma(s, -1.0 / 6227020800.0, 1.0 / 39916800.0), -1.0 / 362880.0), 1.0 / 5040.0), -1.0 / 120.0), 1.0 / 6.0);
    const double cs = fma(-s, b, 1.0);                      // cos(theta)
    const double h2 = (1.0 + cs) * 0.5;                     // cos^2(theta / 2) in (0.93, 1]
    const double ich = po_rsqrt(h2), ch = h2 * ich, kq = a * 0.5 * ich;
    const double eq[4] = { kq * u[0], kq * u[1], kq * u[2], ch };
    // V = I + b [w]x + c [w]x^2 applied to upsilon: w x v and w x (w x v)
    const double* w = u; const double* v = u + 3;
    const double wv[3] = { w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0] };
    const double wwv[3] = { w[1] * wv[2] - w[2] * wv[1], w[2] * wv[0] - w[0] * wv[2], w[0] * wv[1] - w[1] * wv[0] };
    double rt[3]; const double told[3] = { t[0], t[1], t[2] }; const double qo[4] = { q[0], q[1], q[2], q[3] };
    quat_rot(eq, told, rt);
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = fma(c, wwv[k], fma(b, wv[k], v[k])) + rt[k];
    double nq[4];
    nq[3] = eq[3] * qo[3] - eq[0] * qo[0] - eq[1] * qo[1] - eq[2] * qo[2];
    nq[0] = eq[3] * qo[0] + eq[0] * qo[3] + eq[1] * qo[2] - eq[2] * qo[1];
    nq[1] = eq[3] * qo[1] + eq[1] * qo[3] + eq[2] * qo[0] - eq[0] * qo[2];
    nq[2] = eq[3] * qo[2] + eq[2] * qo[3] + eq[0] * qo[1] - eq[1] * qo[0];
    double in = po_rsqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    if (nq[3] < 0) in = -in;
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = nq[k] * in;
}

// CACHED: every problem of the launch has at most PO_T * PO_EPT edges -- a thread keeps its edges (point, observation, weight, kind, active flag,
// last chi2) in registers for the whole call, so that the two sweeps of every LM iteration are not two rounds of global-memory latency
// (13 -> 10 us per iteration at 1 750 edges).  Otherwise the edges are re-read from global memory in every sweep.
// (round 3, measured and dropped: ONE full sweep per trial -- chi2 of the trial and, if it is accepted, the system of the next iteration -- instead of a
// full sweep per iteration plus a light one per trial: 216 -> 272 us per call at 1 750 edges; the light sweep is a third of the full one, and rejected
// trials then pay the full one.)
#ifndef PO_EPT
#define PO_EPT 4
#endif
#define PO_FOR_EDGES(j, i) _Pragma("unroll") for (int j = 0; j < (CACHED ? PO_EPT : 1); j++) for (int i = tid + (CACHED ? j * PO_T : 0); i < nE; i += (CACHED ? nE : PO_T))
#define PO_EDGE_XZ(j, i) double eX[3], eZ[3]; _Pragma("unroll") for (int k_ = 0; k_ < 3; k_++) { eX[k_] = CACHED ? (double)cX[j][k_] : PT[3 * (i) + k_]; eZ[k_] = CACHED ? (double)cZ[j][k_] : OBS[3 * (i) + k_]; }
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
    // (the world points, observations and weights of the C-ABI are floats widened on the host: holding them as floats is exact and keeps the kernel out of scratch)
    float cX[CACHED ? PO_EPT : 1][3], cZ[CACHED ? PO_EPT : 1][3], cW[CACHED ? PO_EPT : 1]; double cL[CACHED ? PO_EPT : 1];
    int cD[CACHED ? PO_EPT : 1], cA[CACHED ? PO_EPT : 1];
    PO_FOR_EDGES(j, i) {
        ACT[i] = 1;
        if (CACHED) {
#pragma unroll
            for (int k = 0; k < 3; k++) { cX[j][k] = (float)PT[3 * i + k]; cZ[j][k] = (float)OBS[3 * i + k]; }
            cW[j] = (float)W[i]; cD[j] = DIM[i]; cA[j] = 1; cL[j] = 0.0;
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
                const PoRt P = po_rt(s_pose);
                PO_FOR_EDGES(j, i) {
                    if (!(CACHED ? cA[j] : (int)ACT[i])) continue;
                    const int D = CACHED ? cD[j] : (int)DIM[i];
                    const double wi = CACHED ? (double)cW[j] : W[i];
                    PO_EDGE_XZ(j, i);
                    double err[3], Xc[3], rho[2] = { 0.0, 1.0 };
                    const double chi = po_edge_error(P, eX, eZ, wi, D, fx, fy, cx, cy, bf, err, Xc);
                    if (CACHED) cL[j] = chi; else LAST[i] = chi;
                    double wgt = wi;
                    if (robust) { huber(chi, D == 2 ? d2 : d3, rho); acc[27] += rho[0]; wgt *= rho[1]; }
                    else acc[27] += chi;
                    // d e / d pose (types_six_dof_expmap.cpp:118-131, 214-233) through one reciprocal; columns 4 of row 0 and 3 of rows 1, 2 are zero
                    const double iz = 1.0 / Xc[2], xz = Xc[0] * iz, yz = Xc[1] * iz;
                    const double fxz = fx * iz, fyz = fy * iz;
                    const double J0[5] = { xz * yz * fx, -fma(xz, xz, 1.0) * fx, yz * fx, -fxz, xz * fxz };          // columns 0 1 2 3 5
                    const double J1[5] = { fma(yz, yz, 1.0) * fy, -xz * yz * fy, -xz * fy, -fyz, yz * fyz };         // columns 0 1 2 4 5
                    const double bz = (D == 3) ? bf * iz : 0.0, s3 = (D == 3) ? 1.0 : 0.0;
                    const double J2[5] = { s3 * J0[0] - bz * yz, fma(bz, xz, s3 * J0[1]), s3 * J0[2], s3 * J0[3], s3 * J0[4] - bz * iz };   // columns 0 1 2 3 5
                    // one row of the Jacobian at a time (its 5 non-zero entries against themselves: 15 products, and the 5 of b): the sums of all three
                    // rows in one expression kept 30 doubles live next to the 28 sums and the cached edges, and the kernel spilled inside this loop
                    po_row<0>(J0, wgt, -wgt * err[0], acc);
                    __builtin_amdgcn_sched_barrier(0);
                    po_row<0>(J2, wgt, -wgt * err[2], acc);
                    __builtin_amdgcn_sched_barrier(0);
                    po_row<1>(J1, wgt, -wgt * err[1], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            {
                const double tot = wave_transpose_reduce32(acc);
                const int lane = tid & 63;
                __syncthreads();
                if (lane < 32) s_w[tid >> 6][__brev((unsigned)lane) >> 27] = tot;      // id(l): the five low bits of the lane, reversed
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
                    double q[4] = { s_pose[0], s_pose[1], s_pose[2], s_pose[3] }, t[3] = { s_pose[4], s_pose[5], s_pose[6] };
                    po_oplus(xx, q, t);                                                             // oplus
                    for (int a = 0; a < 4; a++) s_pose[a] = q[a];
                    for (int a = 0; a < 3; a++) s_pose[4 + a] = t[a];
                    double scale = 0;
                    for (int a = 0; a < 6; a++) scale = fma(xx[a], fma(s_lambda, xx[a], s_tot[21 + a]), scale);   // computeScale (:182-189)
                    s_rho = scale + 1e-3;
                    s_ok2 = ok2;
                }
                __syncthreads();
                double part = 0;
                {
                    const PoRt P = po_rt(s_pose);
                    PO_FOR_EDGES(j, i) {
                        if (!(CACHED ? cA[j] : (int)ACT[i])) continue;
                        const int D = CACHED ? cD[j] : (int)DIM[i];
                        PO_EDGE_XZ(j, i);
                        double err[3], Xc[3], rho[2];
                        double c = po_edge_error(P, eX, eZ, CACHED ? (double)cW[j] : W[i], D, fx, fy, cx, cy, bf, err, Xc);
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
                        const double r21 = 2 * rho_lm - 1;
                        double alpha = 1. - r21 * r21 * r21;                                         // pow(2 rho - 1, 3)
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
            const PoRt P = po_rt(s_pose);
            PO_FOR_EDGES(j, i) {
                const int act = CACHED ? cA[j] : (int)ACT[i];
                const int D = CACHED ? cD[j] : (int)DIM[i];
                double fresh = 0, depth = 1;
                if (need_eval) { PO_EDGE_XZ(j, i); double err[3], Xc[3]; fresh = po_edge_error(P, eX, eZ, CACHED ? (double)cW[j] : W[i], D, fx, fy, cx, cy, bf, err, Xc); depth = Xc[2]; }
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
