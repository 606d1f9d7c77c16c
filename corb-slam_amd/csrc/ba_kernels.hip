// ba_kernels.hip -- global bundle adjustment on gfx950 (FP64).
// Reference arithmetic: g2o EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ (G/types/types_six_dof_expmap.{h,cpp}),
// BaseBinaryEdge::constructQuadraticForm (G/core/base_binary_edge.hpp:55-120), BlockSolver_6_3 Schur
// solve (G/core/block_solver.hpp:354-486), SE3Quat::exp / operator* (G/types/se3quat.h).
//
// Kernels (edges are sorted by landmark; "free" = not fixed):
//   ba_error_kernel      e = z - pi(T X), chi2 / Huber rho                 (computeActiveErrors + activeRobustChi2)
//   ba_linearize_kernel  A = de/dX, B = de/dT, per-edge A'WA, B'WB, B'WA, -A'We, -B'We   (buildSystem)
//   ba_sum_points / ba_sum_poses   ordered (deterministic) block sums into Hll,b_l / Hpp,b_p
//   ba_schur_prepare     Dinv = (Hll + lambda I)^-1, db = Dinv b_l
//   ba_schur_pairs       S(i,j) -= (Hpl_i Dinv) Hpl_j'  -- one wavefront per landmark, 16x16 tiles on
//                        v_mfma_f64_16x16x4_f64 (the K=3 contraction over the landmark axes padded to 4)
//   ba_reduced_rhs       b_schur = b_p - sum_e Hpl_e db
//   ba_backsub           x_l = Dinv (b_l - sum_e Hpl_e' x_p)
//   ba_update            T <- exp(dx) T ; X <- X + dx
#include "ba_internal.h"
#include "ba_multilevel.h"
#include "device_util.h"
#include <cfloat>
#include <algorithm>
#include "ba_math.h"
#include "lane_exchange.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

// error of one edge; returns chi2 = w |e|^2
// (pointer form: the same operations on operands the caller has fetched -- ba_hpp_scratch_kernel holds the keyframe's pose and camera in registers)
__device__ __forceinline__ double edge_error_p(const double* q, const double* t, const double* cam, const double* X, const double* z, double w, int dim, double* err, double* Xc)
{
    quat_rot(q, X, Xc);
    Xc[0] += t[0]; Xc[1] += t[1]; Xc[2] += t[2];
    if (dim == 2) {
        err[0] = z[0] - (Xc[0] / Xc[2] * cam[0] + cam[2]);
        err[1] = z[1] - (Xc[1] / Xc[2] * cam[1] + cam[3]);
        err[2] = 0;
        return w * (err[0] * err[0] + err[1] * err[1]);
    }
    const float invz = (float)(1.0 / Xc[2]);            // `const float invz = 1.0f/trans_xyz[2]` (types_six_dof_expmap.cpp:151)
    const double r0 = Xc[0] * invz * cam[0] + cam[2];
    const double r1 = Xc[1] * invz * cam[1] + cam[3];
    const double r2 = r0 - cam[4] * invz;
    err[0] = z[0] - r0; err[1] = z[1] - r1; err[2] = z[2] - r2;
    return w * (err[0] * err[0] + err[1] * err[1] + err[2] * err[2]);
}
__device__ __forceinline__ double edge_error(const CorbBADev& d, int i, double* err, double* Xc)
{
    const int vp = d.e_vpose[i], vx = d.e_vpoint[i];
    return edge_error_p(d.pose_q + 4 * (size_t)vp, d.pose_t + 3 * (size_t)vp, d.cam + 5 * (size_t)vp, d.pt + 3 * (size_t)vx, d.e_obs + 3 * (size_t)i, d.e_w[i], d.e_dim[i], err, Xc);
}

// _jacobianOplusXj of EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ (types_six_dof_expmap.cpp:88-124, 169-217): rows 0..2 of the 3 x 6 block, row-major in B[0..17]
__device__ __forceinline__ void edge_pose_jacobian(double x, double y, double z, double z_2, double fx, double fy, double bf, int D, double* B)
{
    B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
    B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
    if (D == 3) { B[12] = B[0] - bf * y / z_2; B[13] = B[1] + bf * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf / z_2; }
    else { B[12] = B[13] = B[14] = B[15] = B[16] = B[17] = 0; }
}

// Jacobian blocks with ONE division (see ba_edge_jacobians_fast)
// (the arithmetic: WANT_A / WANT_B select the block a caller needs)
template <bool WANT_A, bool WANT_B>
__device__ __forceinline__ void ba_jacobians_fast_core(const double* R, const double* Xc, const double* cam, int D, double* A, double* B)
{
    const double fx = cam[0], fy = cam[1], bf = cam[4];
    const double x = Xc[0], y = Xc[1], iz = 1.0 / Xc[2], iz2 = iz * iz;
    const double fxz = fx * iz, fyz = fy * iz, fxx = fx * x * iz2, fyy = fy * y * iz2, bz = bf * iz2;
    if (WANT_A) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A[j] = -fxz * R[j] + fxx * R[6 + j];
            A[3 + j] = -fyz * R[3 + j] + fyy * R[6 + j];
            A[6 + j] = D == 3 ? A[j] - bz * R[6 + j] : 0.0;
        }
    }
    if (WANT_B) {
        B[0] = x * y * iz2 * fx; B[1] = -(1 + x * x * iz2) * fx; B[2] = y * iz * fx; B[3] = -fxz; B[4] = 0; B[5] = fxx;
        B[6] = (1 + y * y * iz2) * fy; B[7] = -x * y * iz2 * fy; B[8] = -x * iz * fy; B[9] = 0; B[10] = -fyz; B[11] = fyy;
        if (D == 3) { B[12] = B[0] - bz * y; B[13] = B[1] + bz * x; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bz; }
        else { B[12] = B[13] = B[14] = B[15] = B[16] = B[17] = 0; }
    }
}

__device__ __forceinline__ double block_sum_256(double v, double* red)
{
    v = lx_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// The last workgroup of a launch to arrive (a ticket in device memory) sums the launch's per-workgroup partials in index order -- the arithmetic of the
// former stand-alone reduction kernel, one stream operation less per sum.  Partials are published and read with agent-scope atomics and the ticket is
// taken after the store has been acknowledged (no cache-wide fence: see the PCG group reductions below).  Called by every thread of every workgroup
// with the workgroup's sum; writes *out once; the ticket is left at 0.  (Fixed summation order => run-to-run reproducible.)
// One LM trial's decision on the device (BALMCtl): what the host loop of ba_lm_device does with the trial's read-back when the trial is ACCEPTED
// (optimization_algorithm_levenberg.cpp:120-141: rho, lambda *= max(1/3, 1 - (2 rho - 1)^3), ni = 2; then ORB-SLAM2's stop rule); anything else stops the chain.
// scal = the trial's scalars (chi2, -, scale, ...), bad = the two status words, epoch = the trial's number.  One thread.
__device__ __forceinline__ void ba_lm_decide(BALMCtl* c, const double* scal, const int* bad, int epoch)
{
    if (c->stop) return;
    const bool ok2 = !(bad[0] == epoch || bad[1] != 0);
    const double tempChi = ok2 ? scal[0] : DBL_MAX;
    double rho = c->currentChi - tempChi;
    rho /= (ok2 ? scal[2] : 0.0) + 1e-3;
    if (!(rho > 0 && isfinite(tempChi))) { c->stop = 3; return; }
    const double iniChi = c->currentChi;
    double alpha = 1. - pow((2 * rho - 1), 3.0);
    alpha = fmin(alpha, 2. / 3.);
    c->lambda *= fmax(1. / 3., alpha); c->ni = 2; c->currentChi = tempChi;
    const int k = c->it_done;
    c->chi2_hist[k] = tempChi; c->lambda_hist[k] = c->lambda;
    c->it_done = k + 1; c->trials += 1;
    if ((iniChi - tempChi) * 1e3 < iniChi) c->nBad += 1; else c->nBad = 0;
    if (c->nBad >= 3) c->stop = 2;               // (a chain that runs to its end stays at 0: the next iteration's linearisation is enqueued behind this kernel)
}
// ctl != nullptr: the sum is a trial's chi2 (out = the trial's scalars) and the thread that files it takes the trial's decision at once -- the one-thread launch that
// used to follow (ba_lm_ctl_kernel, ~4.5 us of a local window's ~60 us trial) is gone
__device__ __forceinline__ void ba_finish_sum(double s, double* partial, double* out, int* tick, double* red, BALMCtl* ctl = nullptr, const int* ctl_bad = nullptr, int ctl_epoch = 0)
{
    __shared__ int s_last_sum;
    if (gridDim.x == 1) { if (threadIdx.x == 0) { *out = s; if (ctl) ba_lm_decide(ctl, out, ctl_bad, ctl_epoch); } return; }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partial[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last_sum = __hip_atomic_fetch_add(tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last_sum) return;
    double acc = 0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) acc += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double t = block_sum_256(acc, red);
    if (threadIdx.x == 0) { *out = t; *tick = 0; if (ctl) ba_lm_decide(ctl, out, ctl_bad, ctl_epoch); }
}

// chi2 of the active edges (and the per-edge values): per-workgroup partials, summed by the last workgroup
__global__ __launch_bounds__(256) void ba_error_kernel(CorbBADev d, double* partial, double* out, const int* ctl_bad, int ctl_epoch)
{
    __shared__ double red[4];
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    double acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.nE; i += gridDim.x * 256) {
        double err[3], Xc[3], rho[2];
        double c = edge_error(d, i, err, Xc);
        if (d.e_chi2) d.e_chi2[i] = c;
        if (d.robust) { huber(c, d.e_dim[i] == 2 ? d.delta2 : d.delta3, rho); c = rho[0]; }
        acc += c;
    }
    const double s = block_sum_256(acc, red);
    ba_finish_sum(s, partial, out, d.red_tick, red, ctl_bad ? d.ctl : nullptr, ctl_bad, ctl_epoch);
}

__global__ __launch_bounds__(256) void ba_reduce_kernel(const double* partial, int n, double* out)
{
    __shared__ double red[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const double s = block_sum_256(acc, red);
    if (threadIdx.x == 0) *out = s;
}

// linearizeOplus + constructQuadraticForm, one thread per edge; per-edge blocks are summed later
// (body shared by the stand-alone kernel and the fused small-problem kernel: vbid / vtid = virtual block and thread index)
__device__ __forceinline__ void ba_linearize_body(const CorbBADev& d, const int vbid, const int vtid)
{
    const int i = vbid * 256 + vtid;
    if (i >= d.nE) return;
    double err[3], Xc[3], A[9], B[18];
    const double chi = edge_error(d, i, err, Xc);
    const int D = d.e_dim[i];
    double R[9]; quat_to_R(d.pose_q + 4 * (size_t)d.e_vpose[i], R);
    const double x = Xc[0], y = Xc[1], z = Xc[2], z_2 = z * z;
    const double* cam = d.cam + 5 * (size_t)d.e_vpose[i];
    const double fx = cam[0], fy = cam[1], bf = cam[4];
    if (D == 2) {
        const double tmp[6] = { fx, 0, -x / z * fx, 0, fy, -y / z * fy };
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < 3; j++) A[a * 3 + j] = -1. / z * (tmp[a * 3] * R[j] + tmp[a * 3 + 1] * R[3 + j] + tmp[a * 3 + 2] * R[6 + j]);
        A[6] = A[7] = A[8] = 0;
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A[j] = -fx * R[j] / z + fx * x * R[6 + j] / z_2;
            A[3 + j] = -fy * R[3 + j] / z + fy * y * R[6 + j] / z_2;
            A[6 + j] = A[j] - bf * R[6 + j] / z_2;
        }
    }
    B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
    B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
    if (D == 3) { B[12] = B[0] - bf * y / z_2; B[13] = B[1] + bf * x / z_2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf / z_2; }
    else { B[12] = B[13] = B[14] = B[15] = B[16] = B[17] = 0; }
    double w = d.e_w[i];
    if (d.robust) { double rho[2]; huber(chi, D == 2 ? d.delta2 : d.delta3, rho); w *= rho[1]; }   // weightedOmega = rho'(e) Omega
    double* o = d.edge_blk + (size_t)i * BA_EDGE_STRIDE;
    // [0..5] A'WA upper (00 01 02 11 12 22) | [6..8] -A'We | [9..26] JB = (sqrt(w) B)' (6 x 3, row-major) | [27..29] r = -sqrt(w) e.
    // The pose blocks are NOT formed per edge: Hpp = sum_e JB_e JB_e' and b_p = sum_e JB_e r_e are one contraction over (edge, residual row) per keyframe on the
    // FP64 matrix cores (ba_hpp_mfma_kernel).  B'WA (6 x 3, the Hpl block) goes to its own compact array (hpl): the Schur kernels gather these blocks pair by
    // pair, and inside one long record every one of them dragged extra cache lines along
    int k = 0;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int c = a; c < 3; c++) o[k++] = w * (A[a] * A[c] + A[3 + a] * A[3 + c] + A[6 + a] * A[6 + c]);
#pragma unroll
    for (int a = 0; a < 3; a++) o[k++] = -w * (A[a] * err[0] + A[3 + a] * err[1] + A[6 + a] * err[2]);
    const double sw = sqrt(w);
#pragma unroll
    for (int a = 0; a < 6; a++) { o[k++] = sw * B[a]; o[k++] = sw * B[6 + a]; o[k++] = sw * B[12 + a]; }
    o[k++] = -sw * err[0]; o[k++] = -sw * err[1]; o[k++] = -sw * err[2];
    double* hw = d.hpl + (size_t)i * 18;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) hw[a * 3 + c] = w * (B[a] * A[c] + B[6 + a] * A[3 + c] + B[12 + a] * A[6 + c]);
}
__global__ __launch_bounds__(256) void ba_linearize_kernel(CorbBADev d) { ba_linearize_body(d, blockIdx.x, threadIdx.x); }

// Hll, b_l : one thread per free landmark, its edges are contiguous [loff[l], loff[l+1])
__device__ __forceinline__ void ba_sum_points_body(const CorbBADev& d, const int vbid, const int vtid)
{
    const int l = vbid * 256 + vtid;
    if (l >= d.nL) return;
    double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int e = d.loff[l]; e < d.loff[l + 1]; e++) {
        const double* o = d.edge_blk + (size_t)e * BA_EDGE_STRIDE;
#pragma unroll
        for (int k = 0; k < 6; k++) h[k] += o[k];
#pragma unroll
        for (int k = 0; k < 3; k++) g[k] += o[6 + k];
    }
    double* H = d.Hll + 9 * (size_t)l;
    H[0] = h[0]; H[1] = h[1]; H[2] = h[2]; H[3] = h[1]; H[4] = h[3]; H[5] = h[4]; H[6] = h[2]; H[7] = h[4]; H[8] = h[5];
    double* b = d.b + d.sp + 3 * (size_t)l;
    b[0] = g[0]; b[1] = g[1]; b[2] = g[2];
}
__global__ __launch_bounds__(256) void ba_sum_points_kernel(CorbBADev d) { ba_sum_points_body(d, blockIdx.x, threadIdx.x); }

// Hpp, b_p : one wavefront per free pose; lane-strided over the pose's edge list, butterfly sum (the form of the one-workgroup optimiser; the
// stand-alone path uses ba_hpp_mfma_kernel below)
__device__ __forceinline__ void ba_sum_poses_body(const CorbBADev& d, const int vbid, const int vtid)
{
    const int k = vbid * 4 + (vtid >> 6), lane = vtid & 63;
    if (k >= d.nP) return;
    double acc[27];
#pragma unroll
    for (int j = 0; j < 27; j++) acc[j] = 0;
    for (int ii = d.poff[k] + lane; ii < d.poff[k + 1]; ii += 64) {
        const double* o = d.edge_blk + (size_t)d.pedge[ii] * d.edge_stride + d.edge_jb;       // JB (6 x 3) | r (3)
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = a; c < 6; c++) acc[t++] += o[a * 3] * o[c * 3] + o[a * 3 + 1] * o[c * 3 + 1] + o[a * 3 + 2] * o[c * 3 + 2];
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] += o[a * 3] * o[18] + o[a * 3 + 1] * o[19] + o[a * 3 + 2] * o[20];
    }
#pragma unroll
    for (int j = 0; j < 27; j++) {
        double v = acc[j];
        v = lx_wave_sum(v);
        acc[j] = v;
    }
    if (lane == 0) {
        double* H = d.Hpp + 36 * (size_t)k;
        int t = 0;
        for (int a = 0; a < 6; a++) for (int c = a; c < 6; c++) { H[a * 6 + c] = acc[t]; H[c * 6 + a] = acc[t]; t++; }
        for (int a = 0; a < 6; a++) d.b[6 * (size_t)k + a] = acc[21 + a];
    }
}
__global__ __launch_bounds__(256) void ba_sum_poses_kernel(CorbBADev d) { ba_sum_poses_body(d, blockIdx.x, threadIdx.x); }

// J'OmegaJ accumulation of the pose blocks on the FP64 matrix cores (BaseBinaryEdge::constructQuadraticForm, G/core/base_binary_edge.hpp:74-90:
// Hpp += B' Omega B, b_p += -B' Omega e): with JB_e = (sqrt(w) B_e)' (6 x 3) and r_e = -sqrt(w) e_e
//     [Hpp | b_p](keyframe) = sum over its edges of JB_e [JB_e' | r_e]
// is ONE contraction of depth 3 x edges per keyframe.  One wavefront per keyframe, v_mfma_f64_4x4x4_4b_f64 with the contraction split over its four
// blocks exactly as in ba_schur_mfma_kernel: lane (k = lane>>4, blk = (lane>>2)&3, i = lane&3) owns edge 4 blk + k of a group of 16 and feeds rows
// i / 4+i of JB as A and columns i / 4+i of [JB' | r] as B (column 6 = r, column 7 and rows 6, 7 are padding); three instructions (the three
// residual rows) per quadrant.  Fixed order, no atomics: deterministic.  Replaces the 21 + 6 per-edge products and their butterfly sums.
// SPLIT > 1 (a local window: a handful of keyframes with thousands of observations each): one WORKGROUP of SPLIT wavefronts per keyframe, wavefront w
// takes the groups w, w + SPLIT, ...; the partial tiles meet in LDS and are summed in wavefront order (fixed: deterministic).
#define BA_SMALL_SPLIT 16
#define BA_SMALL_SPLIT_MAX_UNITS 128          // keyframes / blocks up to which the split form is launched
template <int SPLIT>
__global__ __launch_bounds__(SPLIT == 1 ? 256 : 64 * SPLIT) void ba_hpp_mfma_kernel(CorbBADev d)
{
    __shared__ double part[SPLIT == 1 ? 1 : SPLIT][64];
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kf = __builtin_amdgcn_readfirstlane(SPLIT == 1 ? blockIdx.x * 4 + wave : blockIdx.x);
    const int g0 = SPLIT == 1 ? 0 : 16 * wave;             // first pair of this wavefront's first group
    constexpr int GS = 16 * SPLIT;                         // distance between a wavefront's groups
    if (kf >= d.nP) return;
    const int i0 = __builtin_amdgcn_readfirstlane(d.poff[kf]), n = __builtin_amdgcn_readfirstlane(d.poff[kf + 1]) - i0;
    const int k = lane >> 4, blk = (lane >> 2) & 3, i4 = lane & 3;
    const int pl = 4 * blk + k;
    // A rows: i4 and 4 + i4 (rows 6, 7 -> row 5 again, discarded); B columns: i4 and 4 + i4 where column 6 is r (offset 18) and column 7 repeats it
    const int alo = i4 * 3, ahi = min(4 + i4, 5) * 3, bhi = (i4 < 2 ? (4 + i4) : 6) * 3;
    double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
    if (n > g0) {
        int e = d.pedge[i0 + min(g0 + pl, n - 1)];
        for (int c0 = g0; c0 < n; c0 += GS) {
            const bool live = c0 + pl < n;
            const double* J = d.edge_blk + (size_t)e * d.edge_stride + d.edge_jb;
            double al[3], ah[3], bh[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { al[c] = J[alo + c]; ah[c] = J[ahi + c]; bh[c] = J[bhi + c]; }
            e = d.pedge[i0 + min(c0 + GS + pl, n - 1)];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double xl = live ? al[c] : 0.0, xh = live ? ah[c] : 0.0;
                a00 = __builtin_amdgcn_mfma_f64_4x4x4f64(xl, al[c], a00, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f64_4x4x4f64(xl, bh[c], a01, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f64_4x4x4f64(xh, al[c], a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f64_4x4x4f64(xh, bh[c], a11, 0, 0, 0);
            }
        }
    }
    a00 += lx_xor<4>(a00); a01 += lx_xor<4>(a01); a10 += lx_xor<4>(a10); a11 += lx_xor<4>(a11);
    a00 += lx_xor<8>(a00); a01 += lx_xor<8>(a01); a10 += lx_xor<8>(a10); a11 += lx_xor<8>(a11);
    double acc = blk == 0 ? a00 : blk == 1 ? a01 : blk == 2 ? a10 : a11;
    if (SPLIT > 1) {
        part[wave][lane] = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = 0;
#pragma unroll
        for (int w = 0; w < SPLIT; w++) acc += part[w][lane];
    }
    const int row = 4 * (blk >> 1) + k, col = 4 * (blk & 1) + i4;           // D[blk][i][j] at lane 16 i + 4 blk + j
    if (row >= 6) return;
    if (col < 6) d.Hpp[36 * (size_t)kf + row * 6 + col] = acc;
    else if (col == 6) d.b[6 * (size_t)kf + row] = acc;
}

// Maps (d.hpp_scratch): the same contraction WITHOUT the JB | r records in memory.  Only this kernel ever read them, 168 bytes per observation gathered in keyframe
// order behind the 4.6 GB the build kernel had written (1.2 + 1.3 ms per trial at 27.5 M observations).  Here the wavefront of a keyframe holds the pose and the
// camera in registers, streams the keyframe's edges from kfrec (40 bytes each, consecutive), gathers the map point, and every lane forms its edge's record --
// the error by the reference's expressions, the 3 x 6 block with one division (ba_jacobians_fast_core; with the reference's fourteen divisions per block the kernel
// reproduced the gather form's Hpp and b_p bit for bit, at 0.63 instead of 0.52 ms) -- into the wavefront's LDS; the matrix instructions then read the groups of 16
// edges from there in the old order.
__global__ __launch_bounds__(256) void ba_kfrec_kernel(CorbBADev d, BAKfRec* out, int n)
{
    const int ii = blockIdx.x * 256 + threadIdx.x;
    if (ii >= n) return;
    const int e = d.pedge[ii];
    BAKfRec r; r.obs[0] = d.e_obs[3 * (size_t)e]; r.obs[1] = d.e_obs[3 * (size_t)e + 1]; r.obs[2] = d.e_obs[3 * (size_t)e + 2]; r.w = d.e_w[e]; r.vpoint = d.e_vpoint[e]; r.dim = d.e_dim[e];
    out[ii] = r;
}
void ba_launch_kfrec(const CorbBADev& d, BAKfRec* out, int n, hipStream_t s) { if (n > 0) hipLaunchKernelGGL(ba_kfrec_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, out, n); }
__global__ __launch_bounds__(256) void ba_hpp_scratch_kernel(CorbBADev d)
{
    __shared__ double stage[4][64 * 21];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kf = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (kf >= d.nP) return;
    const int i0 = __builtin_amdgcn_readfirstlane(d.poff[kf]), n = __builtin_amdgcn_readfirstlane(d.poff[kf + 1]) - i0;
    const int k = lane >> 4, blk = (lane >> 2) & 3, i4 = lane & 3;
    const int pl = 4 * blk + k;
    const int alo = i4 * 3, ahi = min(4 + i4, 5) * 3, bhi = (i4 < 2 ? (4 + i4) : 6) * 3;
    double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
    if (n > 0) {
        const int v = d.pose_vertex[kf];
        double q[4], t[3], cam[5];
#pragma unroll
        for (int c = 0; c < 4; c++) q[c] = d.pose_q[4 * (size_t)v + c];
#pragma unroll
        for (int c = 0; c < 3; c++) t[c] = d.pose_t[3 * (size_t)v + c];
#pragma unroll
        for (int c = 0; c < 5; c++) cam[c] = d.cam[5 * (size_t)v + c];
        double* mine = &stage[wave][lane * 21];
        // (the next 64 edges' records and map points are requested before this chunk's arithmetic: two dependent trips per chunk otherwise, with 12 wavefronts per CU)
        BAKfRec rn = d.kfrec[i0 + min(lane, n - 1)];
        double Xn[3] = { d.pt[3 * (size_t)rn.vpoint], d.pt[3 * (size_t)rn.vpoint + 1], d.pt[3 * (size_t)rn.vpoint + 2] };
        for (int c0 = 0; c0 < n; c0 += 64) {
            const BAKfRec r = rn;
            const double X[3] = { Xn[0], Xn[1], Xn[2] };
            if (c0 + 64 < n) {
                rn = d.kfrec[i0 + min(c0 + 64 + lane, n - 1)];
                Xn[0] = d.pt[3 * (size_t)rn.vpoint]; Xn[1] = d.pt[3 * (size_t)rn.vpoint + 1]; Xn[2] = d.pt[3 * (size_t)rn.vpoint + 2];
            }
            double err[3], Xc[3], B[18];
            const double chi = edge_error_p(q, t, cam, X, r.obs, r.w, r.dim, err, Xc);
            ba_jacobians_fast_core<false, true>(nullptr, Xc, cam, r.dim, nullptr, B);
            double w = r.w;
            if (d.robust) { double rho[2]; huber(chi, r.dim == 2 ? d.delta2 : d.delta3, rho); w *= rho[1]; }
            const double sw = sqrt(w);
            {
                int kk = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) { mine[kk++] = sw * B[a]; mine[kk++] = sw * B[6 + a]; mine[kk++] = sw * B[12 + a]; }
                mine[kk++] = -sw * err[0]; mine[kk++] = -sw * err[1]; mine[kk++] = -sw * err[2];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (c0 + 16 * g >= n) break;                        // (wave-uniform)
                const bool live = c0 + 16 * g + pl < n;
                const double* J = &stage[wave][(16 * g + pl) * 21];
                double al[3], ah[3], bh[3];
#pragma unroll
                for (int c = 0; c < 3; c++) { al[c] = J[alo + c]; ah[c] = J[ahi + c]; bh[c] = J[bhi + c]; }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const double xl = live ? al[c] : 0.0, xh = live ? ah[c] : 0.0;
                    a00 = __builtin_amdgcn_mfma_f64_4x4x4f64(xl, al[c], a00, 0, 0, 0);
                    a01 = __builtin_amdgcn_mfma_f64_4x4x4f64(xl, bh[c], a01, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f64_4x4x4f64(xh, al[c], a10, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f64_4x4x4f64(xh, bh[c], a11, 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    a00 += lx_xor<4>(a00); a01 += lx_xor<4>(a01); a10 += lx_xor<4>(a10); a11 += lx_xor<4>(a11);
    a00 += lx_xor<8>(a00); a01 += lx_xor<8>(a01); a10 += lx_xor<8>(a10); a11 += lx_xor<8>(a11);
    const double acc = blk == 0 ? a00 : blk == 1 ? a01 : blk == 2 ? a10 : a11;
    const int row = 4 * (blk >> 1) + k, col = 4 * (blk & 1) + i4;           // D[blk][i][j] at lane 16 i + 4 blk + j
    if (row >= 6) return;
    if (col < 6) d.Hpp[36 * (size_t)kf + row * 6 + col] = acc;
    else if (col == 6) d.b[6 * (size_t)kf + row] = acc;
}

// max |diag(H)| over all free vertices (computeLambdaInit, optimization_algorithm_levenberg.cpp:166-180)
__global__ __launch_bounds__(256) void ba_maxdiag_kernel(CorbBADev d, double* out)
{
    __shared__ double red[4];
    // grid-stride over all diagonal entries; the maximum of non-negative doubles is the maximum of their bit patterns, so the workgroups
    // combine with an integer atomicMax (exact, order-independent); *out is zeroed by a memset node before the launch
    double m = 0;
    const int T = gridDim.x * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    for (int i = t0; i < d.nP * 6; i += T) m = fmax(m, fabs(d.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)]));
    for (int i = t0; i < d.nL * 3; i += T) m = fmax(m, fabs(d.Hll[9 * (size_t)(i / 3) + 4 * (i % 3)]));
    m = lx_wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(fmax(fmax(red[0], red[1]), fmax(red[2], red[3]))));
}

// S = blockdiag(Hpp + lambda I)   (S is dense sp x sp, zeroed by a memset node before this kernel)
__device__ __forceinline__ void ba_s_diag_body(const CorbBADev& d, const int vbid, const int vtid, double lambda)
{
    const int i = vbid * 256 + vtid;
    if (i >= d.nP * 36) return;
    const int k = i / 36, a = (i % 36) / 6, c = i % 6;
    d.S[(size_t)(6 * k + a) * d.sp + 6 * k + c] = d.Hpp[i] + (a == c ? lambda : 0.0);
}

// Dinv = (Hll + lambda I)^-1 (cofactor inverse as Eigen's Matrix3d::inverse), db = Dinv b_l
__device__ __forceinline__ void ba_schur_prepare_body(const CorbBADev& d, const int vbid, const int vtid, double lambda, int* bad, int epoch)
{
    const int l = vbid * 256 + vtid;
    if (l >= d.nL) return;
    double m[9];
#pragma unroll
    for (int k = 0; k < 9; k++) m[k] = d.Hll[9 * (size_t)l + k];
    m[0] += lambda; m[4] += lambda; m[8] += lambda;
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    if (!isfinite(id)) *bad = epoch;          // (the trial's number: the flag is never cleared, the host compares)
    double* o = d.Dinv + 9 * (size_t)l;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    const double* bl = d.b + d.sp + 3 * (size_t)l;
    double* db = d.db + 3 * (size_t)l;
#pragma unroll
    for (int a = 0; a < 3; a++) db[a] = o[a * 3] * bl[0] + o[a * 3 + 1] * bl[1] + o[a * 3 + 2] * bl[2];
}
__global__ __launch_bounds__(256) void ba_schur_prepare_kernel(CorbBADev d, double lambda, int* bad, int epoch) { ba_schur_prepare_body(d, blockIdx.x, threadIdx.x, lambda, bad, epoch); }


// ------------------------------------------------------------------------------------------------
// Lean build (multi-kernel path, round 3).  Round 2 materialised per edge A'WA | -A'We | JB | r (240 B) and the Hpl block B'WA (144 B) and read them back in
// ba_sum_points, ba_v, ba_reduced_rhs and ba_backsub: 10.5 GB written and ~13 GB re-read per LM iteration at 27.5 M observations (profiles/r03_ba50k_a).
// Now the thread that owns a landmark linearises the landmark's edges itself (they are contiguous: the edges are sorted by landmark), keeps the 3 x 3 sums
// in registers and writes per edge only what another OWNER needs -- JB | r for the keyframe blocks (168 B); per LM trial the same thread re-derives A and B of
// its edges from the estimates (a few hundred flops against 144 B of traffic) and emits V_e = W_e C_l.  With C_l = L^-T (Hll + lambda I = L L') and
// g_l = C_l' b_l:   W_e Dinv b_l = V_e g_l   and   x_l = Dinv (b_l - sum W_e' x_p) = C_l (g_l - sum V_e' x_p),
// so the reduced right-hand side and the back substitution read V, which the Schur products need anyway, and the Hpl array is gone.
__device__ __forceinline__ double ba_edge_jacobians(const CorbBADev& d, int i, double* err, double* A, double* B, double& w)      // returns the edge's chi2
{
    double Xc[3];
    const double chi = edge_error(d, i, err, Xc);
    const int D = d.e_dim[i];
    double R[9]; quat_to_R(d.pose_q + 4 * (size_t)d.e_vpose[i], R);
    const double x = Xc[0], y = Xc[1], z = Xc[2], z_2 = z * z;
    const double* cam = d.cam + 5 * (size_t)d.e_vpose[i];
    const double fx = cam[0], fy = cam[1], bf = cam[4];
    if (D == 2) {
        const double tmp[6] = { fx, 0, -x / z * fx, 0, fy, -y / z * fy };
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int j = 0; j < 3; j++) A[a * 3 + j] = -1. / z * (tmp[a * 3] * R[j] + tmp[a * 3 + 1] * R[3 + j] + tmp[a * 3 + 2] * R[6 + j]);
        A[6] = A[7] = A[8] = 0;
    } else {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            A[j] = -fx * R[j] / z + fx * x * R[6 + j] / z_2;
            A[3 + j] = -fy * R[3 + j] / z + fy * y * R[6 + j] / z_2;
            A[6 + j] = A[j] - bf * R[6 + j] / z_2;
        }
    }
    edge_pose_jacobian(x, y, z, z_2, fx, fy, bf, D, B);
    w = d.e_w[i];
    if (d.robust) { double rho[2]; huber(chi, D == 2 ? d.delta2 : d.delta3, rho); w *= rho[1]; }   // weightedOmega = rho'(e) Omega
    return chi;
}
// A, B and the weight of one edge for the V blocks: the same quantities as ba_edge_jacobians with ONE division (1 / z; the reference's expressions divide
// fifteen times, and FP64 division is a ~30-instruction sequence).  V is an intermediate of the Schur complement, not a quantity g2o rounds in a particular order;
// the error (two more divisions) is evaluated only when the robust kernel needs chi2 for its weight.
__device__ __forceinline__ void ba_edge_jacobians_fast(const CorbBADev& d, int i, double* A, double* B, double& w)
{
    const int vp = d.e_vpose[i], vx = d.e_vpoint[i];
    double Xc[3], R[9];
    quat_rot(d.pose_q + 4 * (size_t)vp, d.pt + 3 * (size_t)vx, Xc);
    Xc[0] += d.pose_t[3 * (size_t)vp]; Xc[1] += d.pose_t[3 * (size_t)vp + 1]; Xc[2] += d.pose_t[3 * (size_t)vp + 2];
    quat_to_R(d.pose_q + 4 * (size_t)vp, R);
    const int D = d.e_dim[i];
    ba_jacobians_fast_core<true, true>(R, Xc, d.cam + 5 * (size_t)vp, D, A, B);
    w = d.e_w[i];
    if (d.robust) { double err[3], Xe[3], rho[2]; const double chi = edge_error(d, i, err, Xe); huber(chi, D == 2 ? d.delta2 : d.delta3, rho); w *= rho[1]; }
}
// The linearisation of a map (d.hpp_scratch: no JB | r records, so no B here): the error by the reference's expressions (chi2 is an observable), the 3 x 3 block A with
// ONE division like the V blocks' -- Hll and b_l are intermediates of the normal equations like V; the reference's expressions for A divide eighteen times.
__device__ __forceinline__ double ba_edge_linearize_point(const CorbBADev& d, int i, double* err, double* A, double& w)      // returns the edge's chi2
{
    double Xc[3], R[9];
    const double chi = edge_error(d, i, err, Xc);
    const int vp = d.e_vpose[i], D = d.e_dim[i];
    quat_to_R(d.pose_q + 4 * (size_t)vp, R);
    ba_jacobians_fast_core<true, false>(R, Xc, d.cam + 5 * (size_t)vp, D, A, nullptr);
    w = d.e_w[i];
    if (d.robust) { double rho[2]; huber(chi, D == 2 ? d.delta2 : d.delta3, rho); w *= rho[1]; }
    return chi;
}
__device__ __forceinline__ void ba_write_jb(const CorbBADev& d, int i, const double* err, const double* B, double w)
{
    double* o = d.edge_blk + (size_t)i * d.edge_stride + d.edge_jb;
    const double sw = sqrt(w);
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) { o[k++] = sw * B[a]; o[k++] = sw * B[6 + a]; o[k++] = sw * B[12 + a]; }
    o[k++] = -sw * err[0]; o[k++] = -sw * err[1]; o[k++] = -sw * err[2];
}
// Workgroup b < ceil(nL / 256): the free landmarks [256 b, 256 b + 256) and their edges [loff[first], loff[last + 1]), 256 edges at a time -- a thread
// per EDGE computes the Jacobians, writes JB | r and leaves the edge's terms of Hll and b_l in LDS, then a thread per LANDMARK adds its edges' terms in
// edge order (the same sums in the same order as a thread walking its landmark's edges, which is what this kernel did until round 3: with ~5.5 edges
// per landmark the lanes of a wavefront then read and wrote every array at a stride of 5.5 elements, the lines came back once per loop trip -- the
// working set of an XCD's wavefronts is twice its L2 -- and the kernel moved 10 GB per launch for 6 GB of operands: 3.4 ms at 27.5 M observations).
// Workgroups beyond: one edge of a fixed landmark per thread.
// lpb = free landmarks per workgroup: 256 on maps, 32 on local windows (2 000 landmarks in 8 workgroups left 248 CUs idle: 27 us per launch)
// chi_partial != nullptr: the launch is also computeActiveErrors() of the estimates it linearises -- every edge's chi2 (robustified where the kernel is on) is
// filed in e_chi2 and summed (per-workgroup partials, finished by the last workgroup) into *chi_out.  The LM loop linearises a trial's estimates BEFORE it knows
// whether the trial is accepted (it nearly always is) and takes the trial's chi2 from this launch: the separate error pass (0.8 ms per trial at 27.5 M
// observations) is gone; a rejected trial restores the estimates and linearises them again.
__global__ __launch_bounds__(256) void ba_build_lean_kernel(CorbBADev d, int lpb, double* chi_partial, double* chi_out, const int* ctl_bad, int ctl_epoch)
{
    // LDS: per wavefront 64 x 21 doubles.  First the wavefront's JB | r records on their way out (64 records = 10.5 KB of consecutive memory, stored with
    // consecutive lanes on consecutive doubles: see ba_v_lean_kernel), then -- in the same space -- its edges' 9 terms of Hll and b_l for the landmark threads.
    __shared__ double stage[4][64 * 21];
    __shared__ double red[4];
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    const int nLb = (d.nL + lpb - 1) / lpb, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    double chi_acc = 0;
    auto file_chi = [&](int i, double chi) {
        if (!chi_partial) return;
        if (d.e_chi2) d.e_chi2[i] = chi;
        if (d.robust) { double rho[2]; huber(chi, d.e_dim[i] == 2 ? d.delta2 : d.delta3, rho); chi = rho[0]; }
        chi_acc += chi;
    };
#define SH(tt) (&stage[(tt) >> 6][((tt) & 63) * 9])
    if ((int)blockIdx.x >= nLb) {
        const int i = d.loff[d.nL] + ((int)blockIdx.x - nLb) * 256 + t;
        if (i < d.nE) {
            double err[3], A[9], B[18], w;
            double chi;
            if (d.hpp_scratch) chi = ba_edge_linearize_point(d, i, err, A, w);      // (an edge of a fixed landmark: only its chi2 is needed here)
            else { chi = ba_edge_jacobians(d, i, err, A, B, w); ba_write_jb(d, i, err, B, w); }
            file_chi(i, chi);
        }
    } else {
    const int L0 = blockIdx.x * lpb, L1 = min(L0 + lpb, d.nL), l = L0 + t;
    const int e0 = d.loff[L0], e1 = d.loff[L1];
    const int my0 = l < L1 ? d.loff[l] : e1, my1 = l < L1 ? d.loff[l + 1] : e1;
    double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int c0 = e0; c0 < e1; c0 += 256) {
        const int i = c0 + t;
        double hg[9];
        if (i < e1) {
            double err[3], A[9], B[18], w;
            const double chi = d.hpp_scratch ? ba_edge_linearize_point(d, i, err, A, w) : ba_edge_jacobians(d, i, err, A, B, w);
            file_chi(i, chi);
            if (!d.hpp_scratch) {   // the record of ba_write_jb, into the wavefront's stage
                double* o = &stage[wv][lane * 21];
                const double sw = sqrt(w);
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) { o[k++] = sw * B[a]; o[k++] = sw * B[6 + a]; o[k++] = sw * B[12 + a]; }
                o[k++] = -sw * err[0]; o[k++] = -sw * err[1]; o[k++] = -sw * err[2];
            }
            int k = 0;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int c = a; c < 3; c++) hg[k++] = w * (A[a] * A[c] + A[3 + a] * A[3 + c] + A[6 + a] * A[6 + c]);
#pragma unroll
            for (int a = 0; a < 3; a++) hg[6 + a] = -w * (A[a] * err[0] + A[3 + a] * err[1] + A[6 + a] * err[2]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!d.hpp_scratch) {
            const int first = c0 + 64 * wv, nrec = min(64, e1 - first);          // this wavefront's records: edges [first, first + nrec)
            if (nrec > 0) {
                double* o = d.edge_blk + (size_t)first * 21;
#pragma unroll
                for (int k = 0; k < 21; k++) { const int m = k * 64 + lane; if (m < nrec * 21) o[m] = stage[wv][m]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (i < e1) {
#pragma unroll
            for (int k = 0; k < 9; k++) SH(t)[k] = hg[k];
        }
        __syncthreads();
        for (int i2 = max(my0, c0), ie = min(my1, c0 + 256); i2 < ie; i2++) {
            const double* v = SH(i2 - c0);
#pragma unroll
            for (int k = 0; k < 6; k++) h[k] += v[k];
#pragma unroll
            for (int a = 0; a < 3; a++) g[a] += v[6 + a];
        }
        __syncthreads();
    }
    if (l < L1) {
    double* H = d.Hll + 9 * (size_t)l;
    H[0] = h[0]; H[1] = h[1]; H[2] = h[2]; H[3] = h[1]; H[4] = h[3]; H[5] = h[4]; H[6] = h[2]; H[7] = h[4]; H[8] = h[5];
    double* b = d.b + d.sp + 3 * (size_t)l;
    b[0] = g[0]; b[1] = g[1]; b[2] = g[2];
    }
    }
#undef SH
    if (chi_partial) {                                       // (kernel argument: uniform)
        const double sum = block_sum_256(chi_acc, red);
        ba_finish_sum(sum, chi_partial, chi_out, d.red_tick, red, ctl_bad ? d.ctl : nullptr, ctl_bad, ctl_epoch);
    }
}
// per LM trial, thread per edge of a free landmark (adjacent threads write adjacent 144-byte V blocks; a thread per LANDMARK walking its edges measured
// 2.7 ms per trial at 27.5 M observations against 1.6 for round 2's kernel): L L' = Hll + lambda I and C = L^-T by every thread of the landmark (30 flops), the
// landmark's first edge files C and g = C' b_l; V_e = W_e C with W_e = B' (w A) re-derived from the estimates (they are the linearisation point: a rejected
// trial restores them before the next trial's launch)
__global__ __launch_bounds__(256) void ba_v_lean_kernel(CorbBADev d, double lambda, int* bad, int epoch)
{
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    if (d.ctl) lambda = d.ctl->lambda;
    const int i_raw = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i_raw < d.nfree_edges;               // (no early exit: every lane of the wavefront takes part in the staged store below)
    const int i = valid ? i_raw : d.nfree_edges - 1;
    const int l = d.e_point[i];
    const double* H = d.Hll + 9 * (size_t)l;
    const double m00 = H[0] + lambda, m10 = H[3], m11 = H[4] + lambda, m20 = H[6], m21 = H[7], m22 = H[8] + lambda;
    const double l00 = sqrt(m00), i00 = 1.0 / l00;
    const double l10 = m10 * i00, l20 = m20 * i00;
    const double d11 = m11 - l10 * l10, l11 = sqrt(d11), i11 = 1.0 / l11;
    const double l21 = (m21 - l20 * l10) * i11;
    const double d22 = m22 - l20 * l20 - l21 * l21, l22 = sqrt(d22), i22 = 1.0 / l22;
    // C = L^-T: c00 = 1/l00, c01 = -l10 c00 / l11, c11 = 1/l11, c02 = -(l20 c00 + l21 c01) / l22, c12 = -l21 c11 / l22, c22 = 1/l22
    const double c00 = i00, c11 = i11, c22 = i22;
    const double c01 = -l10 * c00 * i11, c12 = -l21 * c11 * i22, c02 = -(l20 * c00 + l21 * c01) * i22;
    if (valid && i == d.loff[l]) {
        if (!(m00 > 0) || !(d11 > 0) || !(d22 > 0) || !isfinite(i00 * i11 * i22)) *bad = epoch;
        double* Co = d.Dinv + 9 * (size_t)l;
        Co[0] = c00; Co[1] = c01; Co[2] = c02; Co[3] = c11; Co[4] = c12; Co[5] = c22;
        const double* bl = d.b + d.sp + 3 * (size_t)l;
        double* g = d.db + 3 * (size_t)l;
        g[0] = c00 * bl[0]; g[1] = c01 * bl[0] + c11 * bl[1]; g[2] = c02 * bl[0] + c12 * bl[1] + c22 * bl[2];      // C' b
    }
    // The 144-byte blocks leave through LDS: a wavefront's 64 blocks are 9 KB of consecutive memory, and stored block by block (16 bytes per lane at a
    // lane stride of 144) the stream writes at 3.7 TB/s where consecutive lanes on consecutive 16 bytes reach 6.2 (tools/ubench/stream_patterns.hip,
    // modes 9 / 10).  An edge to a fixed keyframe carries no Schur term: its block is not written.
    __shared__ double2 stage[4][64 * 9];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool wr = valid && d.e_pose[i] >= 0;
    if (wr) {
        double A[9], B[18], w;
        ba_edge_jacobians_fast(d, i, A, B, w);
        // M = w A C (3 x 3), V = B' M (6 x 3)
        double M[9];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const double a0 = w * A[r * 3], a1 = w * A[r * 3 + 1], a2 = w * A[r * 3 + 2];
            M[r * 3] = a0 * c00; M[r * 3 + 1] = a0 * c01 + a1 * c11; M[r * 3 + 2] = a0 * c02 + a1 * c12 + a2 * c22;
        }
        double v[18];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 3; c++) v[a * 3 + c] = B[a] * M[c] + B[6 + a] * M[3 + c] + B[12 + a] * M[6 + c];
#pragma unroll
        for (int k = 0; k < 9; k++) stage[wv][lane * 9 + k] = make_double2(v[2 * k], v[2 * k + 1]);
    }
    const unsigned long long mask = __ballot(wr);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double2* o = reinterpret_cast<double2*>(d.bd) + (size_t)(i_raw - lane) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int m = k * 64 + lane;
        if ((mask >> (m / 9)) & 1ull) o[m] = stage[wv][m];
    }
}
// ---- Round 6: the V blocks in KEYFRAME-LIST order (d.v_kf; maps solved by the row-owner stream kernel) ----
// In landmark (edge) order a keyframe's blocks are scattered -- one per landmark, ~800 bytes apart -- and the row kernel's staging of a range's first operands was a
// gather of 144-byte blocks at two 128-byte lines each: 7.1 GB of lines per launch at the fabric's rate, 0.9 of the kernel's 2.66 ms (profiles/HISTORY_r6.md).  In list
// order the range's blocks are ONE contiguous stretch, and the second operands of a block (p, q) -- V(q, l) over the landmarks both keyframes see, ascending -- are a
// subsequence of q's stretch instead of one block per landmark neighbourhood.  The blocks are written by a thread per LIST ENTRY (consecutive threads, consecutive
// blocks); C_l = L^-T of Hll + lambda I and g_l = C_l' b_l come from a thread per landmark first (the edge-order kernel has every edge of a landmark redo that
// factorisation: 3 square roots and 3 divisions per observation).
__global__ __launch_bounds__(256) void ba_c_kernel(CorbBADev d, double lambda, int* bad, int epoch)
{
    if (d.ctl && d.ctl->stop) return;
    if (d.ctl) lambda = d.ctl->lambda;
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= d.nL) return;
    const double* H = d.Hll + 9 * (size_t)l;
    const double m00 = H[0] + lambda, m10 = H[3], m11 = H[4] + lambda, m20 = H[6], m21 = H[7], m22 = H[8] + lambda;
    const double l00 = sqrt(m00), i00 = 1.0 / l00;
    const double l10 = m10 * i00, l20 = m20 * i00;
    const double d11 = m11 - l10 * l10, l11 = sqrt(d11), i11 = 1.0 / l11;
    const double l21 = (m21 - l20 * l10) * i11;
    const double d22 = m22 - l20 * l20 - l21 * l21, l22 = sqrt(d22), i22 = 1.0 / l22;
    const double c00 = i00, c11 = i11, c22 = i22;
    const double c01 = -l10 * c00 * i11, c12 = -l21 * c11 * i22, c02 = -(l20 * c00 + l21 * c01) * i22;      // (the expressions of ba_v_lean_kernel: the same bits)
    if (!(m00 > 0) || !(d11 > 0) || !(d22 > 0) || !isfinite(i00 * i11 * i22)) *bad = epoch;
    double* Co = d.Dinv + 9 * (size_t)l;
    Co[0] = c00; Co[1] = c01; Co[2] = c02; Co[3] = c11; Co[4] = c12; Co[5] = c22;
    const double* bl = d.b + d.sp + 3 * (size_t)l;
    double* g = d.db + 3 * (size_t)l;
    g[0] = c00 * bl[0]; g[1] = c01 * bl[0] + c11 * bl[1]; g[2] = c02 * bl[0] + c12 * bl[1] + c22 * bl[2];
}
// A wavefront per keyframe (like ba_hpp_scratch_kernel): pose, camera and rotation matrix once, the list's 40-byte kfrec records -- the observations' static data in
// LIST order: one coalesced load per 64 entries where the edge-order arrays would be five scattered ones per entry (the first version of this kernel: 1.47 ms) --,
// the map point and C_l gathered per entry, the 64 blocks of a chunk through LDS into 9 KB of consecutive memory.
__global__ __launch_bounds__(256) void ba_v_kf_kernel(CorbBADev d, int n_list)
{
    __shared__ double2 stage[4][64 * 9];
    if (d.ctl && d.ctl->stop) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kf = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
    if (kf >= d.nP) return;
    const int i0 = __builtin_amdgcn_readfirstlane(d.poff[kf]), n = __builtin_amdgcn_readfirstlane(d.poff[kf + 1]) - i0;
    if (n <= 0) return;
    const int v = d.pose_vertex[kf];
    double q[4], t[3], cam[5], R[9];
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = d.pose_q[4 * (size_t)v + c];
#pragma unroll
    for (int c = 0; c < 3; c++) t[c] = d.pose_t[3 * (size_t)v + c];
#pragma unroll
    for (int c = 0; c < 5; c++) cam[c] = d.cam[5 * (size_t)v + c];
    quat_to_R(q, R);
    // (the next chunk's records, landmarks, map points and C_l are requested before this chunk's arithmetic)
    BAKfRec rn = d.kfrec[i0 + min(lane, n - 1)];
    int ln = d.plm[i0 + min(lane, n - 1)];
    double Xn[3] = { d.pt[3 * (size_t)rn.vpoint], d.pt[3 * (size_t)rn.vpoint + 1], d.pt[3 * (size_t)rn.vpoint + 2] };
    double Cn[6];
#pragma unroll
    for (int c = 0; c < 6; c++) Cn[c] = ln >= 0 ? d.Dinv[9 * (size_t)ln + c] : 0.0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const BAKfRec r = rn; const int l = ln;
        const double X[3] = { Xn[0], Xn[1], Xn[2] };
        const double c00 = Cn[0], c01 = Cn[1], c02 = Cn[2], c11 = Cn[3], c12 = Cn[4], c22 = Cn[5];
        if (c0 + 64 < n) {
            const int ii = i0 + min(c0 + 64 + lane, n - 1);
            rn = d.kfrec[ii]; ln = d.plm[ii];
            Xn[0] = d.pt[3 * (size_t)rn.vpoint]; Xn[1] = d.pt[3 * (size_t)rn.vpoint + 1]; Xn[2] = d.pt[3 * (size_t)rn.vpoint + 2];
#pragma unroll
            for (int c = 0; c < 6; c++) Cn[c] = ln >= 0 ? d.Dinv[9 * (size_t)ln + c] : 0.0;
        }
        const bool wr = c0 + lane < n && l >= 0;             // (an observation of a fixed landmark carries no Schur term: its slot stays unwritten)
        if (wr) {
            double Xc[3];
            quat_rot(q, X, Xc);
            Xc[0] += t[0]; Xc[1] += t[1]; Xc[2] += t[2];
            double A[9], B[18];
            ba_jacobians_fast_core<true, true>(R, Xc, cam, r.dim, A, B);
            double w = r.w;
            if (d.robust) { double err[3], Xe[3], rho[2]; const double chi = edge_error_p(q, t, cam, X, r.obs, r.w, r.dim, err, Xe); huber(chi, r.dim == 2 ? d.delta2 : d.delta3, rho); w *= rho[1]; }
            double M[9];
#pragma unroll
            for (int rr = 0; rr < 3; rr++) {
                const double a0 = w * A[rr * 3], a1 = w * A[rr * 3 + 1], a2 = w * A[rr * 3 + 2];
                M[rr * 3] = a0 * c00; M[rr * 3 + 1] = a0 * c01 + a1 * c11; M[rr * 3 + 2] = a0 * c02 + a1 * c12 + a2 * c22;
            }
            double vv[18];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 3; c++) vv[a * 3 + c] = B[a] * M[c] + B[6 + a] * M[3 + c] + B[12 + a] * M[6 + c];
#pragma unroll
            for (int k = 0; k < 9; k++) stage[wv][lane * 9 + k] = make_double2(vv[2 * k], vv[2 * k + 1]);
        }
        const unsigned long long mask = __ballot(wr);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 8-byte stores: a 2 GiB stream runs at 5.3 TB/s with them and 4.1 with 16-byte ones (tools/ubench/fetch_calib.hip); this kernel 1.48 -> 1.31 ms
        double* o = d.bd + (size_t)(i0 + c0) * 18; const double* sg = reinterpret_cast<const double*>(stage[wv]);
#pragma unroll
        for (int k = 0; k < 18; k++) {
            const int m = k * 64 + lane;
            if ((mask >> (m / 18)) & 1ull) o[m] = sg[m];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
// vslot[e] = the list position (= V block) of edge e; -1 for an edge no free keyframe's list holds
__global__ __launch_bounds__(256) void ba_vslot_kernel(CorbBADev d, int* vslot, int n_list)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_list) vslot[d.pedge[i]] = i;
}
void ba_launch_vslot(const CorbBADev& d, int* vslot, int n_edges, int n_list, hipStream_t s)
{
    (void)hipMemsetAsync(vslot, 0xFF, sizeof(int) * (size_t)(n_edges > 0 ? n_edges : 1), s);
    if (n_list > 0) hipLaunchKernelGGL(ba_vslot_kernel, dim3((n_list + 255) / 256), dim3(256), 0, s, d, vslot, n_list);
}
// b_schur = b_p - sum over the keyframe's edges of V_e g_l   (ordered sum, one wave per keyframe; SPLIT: a workgroup of 16 wavefronts per keyframe for local windows)
template <int SPLIT>
__device__ __forceinline__ void ba_reduced_rhs_lean_body(const CorbBADev& d, const int bid, double (*part)[6])
{
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = SPLIT == 1 ? bid * 4 + wave : bid;
    if (k >= d.nP) return;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int ii = d.poff[k] + (SPLIT == 1 ? lane : tid); ii < d.poff[k + 1]; ii += (SPLIT == 1 ? 64 : 1024)) {
        const int e = d.pedge[ii];
        const int l = d.e_point[e];
        if (l < 0) continue;
        const double* V = d.bd + (size_t)(d.v_kf ? ii : e) * 18;      // (keyframe-list order: the block of list entry ii)
        const double* g = d.db + 3 * (size_t)l;
#pragma unroll
        for (int a = 0; a < 6; a++) acc[a] += V[a * 3] * g[0] + V[a * 3 + 1] * g[1] + V[a * 3 + 2] * g[2];
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        double v = acc[a];
        v = lx_wave_sum(v);
        if (SPLIT == 1) { if (lane == 0) d.x[6 * (size_t)k + a] = d.b[6 * (size_t)k + a] - v; }
        else if (lane == 0) part[wave][a] = v;
    }
    if (SPLIT > 1) {
        __syncthreads();
        if (tid < 6) {
            double v = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) v += part[w][tid];
            d.x[6 * (size_t)k + tid] = d.b[6 * (size_t)k + tid] - v;
        }
    }
}
template <int SPLIT>
__global__ __launch_bounds__(SPLIT == 1 ? 256 : 1024) void ba_reduced_rhs_lean_kernel(CorbBADev d)
{
    __shared__ double part[16][6];
    ba_reduced_rhs_lean_body<SPLIT>(d, (int)blockIdx.x, part);
}
// x_l = C_l (g_l - sum_e V_e' x_p)
// rederive: only where no other thread of the launch changes the estimates (the stand-alone kernel; ba_update_scale_kernel applies the update in the same launch and
// keeps the V blocks)
__device__ __forceinline__ void ba_backsub_lean_one(const CorbBADev& d, const int l, const bool rederive)
{
    double cl[3] = { d.db[3 * (size_t)l], d.db[3 * (size_t)l + 1], d.db[3 * (size_t)l + 2] };
    const int e0 = d.loff[l], nf = d.lnfree[l];
    if (rederive) {
        // Round 6: V_e' x_p = C' A_e' w_e (B_e x_p) with A_e, B_e re-derived from the estimates (still the linearisation point: the update follows this kernel) like
        // ba_v_lean_kernel derives them -- 21 bytes of the edge's index data + L2-resident poses instead of the 144-byte V block (4 GB per trial at 27.5 M observations):
        // t = sum_e A_e' w_e (B_e x_p) in edge order, x_l = C (g - C' t)
        double t[3] = { 0, 0, 0 };
        for (int j = 0; j < nf; j++) {
            const int e = e0 + j;
            double A[9], B[18], w;
            ba_edge_jacobians_fast(d, e, A, B, w);
            const double* xp = d.x + 6 * (size_t)d.e_pose[e];
            double u[3];
#pragma unroll
            for (int r = 0; r < 3; r++) u[r] = w * (B[6 * r] * xp[0] + B[6 * r + 1] * xp[1] + B[6 * r + 2] * xp[2] + B[6 * r + 3] * xp[3] + B[6 * r + 4] * xp[4] + B[6 * r + 5] * xp[5]);
#pragma unroll
            for (int c = 0; c < 3; c++) t[c] += A[c] * u[0] + A[3 + c] * u[1] + A[6 + c] * u[2];
        }
        const double* C = d.Dinv + 9 * (size_t)l;
        cl[0] -= C[0] * t[0]; cl[1] -= C[1] * t[0] + C[3] * t[1]; cl[2] -= C[2] * t[0] + C[4] * t[1] + C[5] * t[2];
        double* xl = d.x + d.sp + 3 * (size_t)l;
        xl[0] = C[0] * cl[0] + C[1] * cl[1] + C[2] * cl[2];
        xl[1] = C[3] * cl[1] + C[4] * cl[2];
        xl[2] = C[5] * cl[2];
        return;
    }
    for (int j = 0; j < nf; j++) {
        const int e = e0 + j;
        const double* V = d.bd + (size_t)(d.v_kf ? d.vslot[e] : e) * 18;
        const double* xp = d.x + 6 * (size_t)d.e_pose[e];
#pragma unroll
        for (int c = 0; c < 3; c++) cl[c] -= V[c] * xp[0] + V[3 + c] * xp[1] + V[6 + c] * xp[2] + V[9 + c] * xp[3] + V[12 + c] * xp[4] + V[15 + c] * xp[5];
    }
    const double* C = d.Dinv + 9 * (size_t)l;
    double* xl = d.x + d.sp + 3 * (size_t)l;
    xl[0] = C[0] * cl[0] + C[1] * cl[1] + C[2] * cl[2];
    xl[1] = C[3] * cl[1] + C[4] * cl[2];
    xl[2] = C[5] * cl[2];
}
__global__ __launch_bounds__(256) void ba_backsub_lean_kernel(CorbBADev d)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l < d.nL) ba_backsub_lean_one(d, l, d.backsub_rederive != 0);
}


// b_schur = b_p - sum over the pose's edges of Hpl_e db(landmark_e)   (ordered sum, one wave per pose)
__device__ __forceinline__ void ba_reduced_rhs_body(const CorbBADev& d, const int vbid, const int vtid)
{
    const int k = vbid * 4 + (vtid >> 6), lane = vtid & 63;
    if (k >= d.nP) return;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int ii = d.poff[k] + lane; ii < d.poff[k + 1]; ii += 64) {
        const int e = d.pedge[ii];
        const int l = d.e_point[e];
        if (l < 0) continue;
        const double* W = d.hpl + (size_t)e * 18;
        const double* db = d.db + 3 * (size_t)l;
#pragma unroll
        for (int a = 0; a < 6; a++) acc[a] += W[a * 3] * db[0] + W[a * 3 + 1] * db[1] + W[a * 3 + 2] * db[2];
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        double v = acc[a];
        v = lx_wave_sum(v);
        if (lane == 0) d.x[6 * (size_t)k + a] = d.b[6 * (size_t)k + a] - v;
    }
}
__global__ __launch_bounds__(256) void ba_reduced_rhs_kernel(CorbBADev d) { ba_reduced_rhs_body(d, blockIdx.x, threadIdx.x); }
// local windows (few keyframes with thousands of observations each): a workgroup of 16 wavefronts per keyframe, wavefront partials summed in wavefront order
__global__ __launch_bounds__(1024) void ba_reduced_rhs_split_kernel(CorbBADev d)
{
    __shared__ double part[16][6];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int ii = d.poff[k] + tid; ii < d.poff[k + 1]; ii += 1024) {
        const int e = d.pedge[ii];
        const int l = d.e_point[e];
        if (l < 0) continue;
        const double* W = d.hpl + (size_t)e * 18;
        const double* db = d.db + 3 * (size_t)l;
#pragma unroll
        for (int a = 0; a < 6; a++) acc[a] += W[a * 3] * db[0] + W[a * 3 + 1] * db[1] + W[a * 3 + 2] * db[2];
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        double v = acc[a];
        v = lx_wave_sum(v);
        if (lane == 0) part[wave][a] = v;
    }
    __syncthreads();
    if (tid < 6) {
        double v = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) v += part[w][tid];
        d.x[6 * (size_t)k + tid] = d.b[6 * (size_t)k + tid] - v;
    }
}
static void ba_launch_reduced_rhs(const CorbBADev& d, hipStream_t s)
{
    if (d.nP <= 0) return;
    if (d.lean) {
        if (d.nP <= 128) hipLaunchKernelGGL(ba_reduced_rhs_lean_kernel<16>, dim3(d.nP), dim3(1024), 0, s, d);
        else hipLaunchKernelGGL(ba_reduced_rhs_lean_kernel<1>, dim3((d.nP + 3) / 4), dim3(256), 0, s, d);
        return;
    }
    if (d.nP <= 128) hipLaunchKernelGGL(ba_reduced_rhs_split_kernel, dim3(d.nP), dim3(1024), 0, s, d);
    else hipLaunchKernelGGL(ba_reduced_rhs_kernel, dim3((d.nP + 3) / 4), dim3(256), 0, s, d);
}

// x_l = Dinv (b_l - sum_e Hpl_e' x_p)
__device__ __forceinline__ void ba_backsub_body(const CorbBADev& d, const int vbid, const int vtid)
{
    const int l = vbid * 256 + vtid;
    if (l >= d.nL) return;
    double cl[3] = { d.b[d.sp + 3 * (size_t)l], d.b[d.sp + 3 * (size_t)l + 1], d.b[d.sp + 3 * (size_t)l + 2] };
    const int e0 = d.loff[l], nf = d.lnfree[l];
    for (int j = 0; j < nf; j++) {
        const int e = e0 + j;
        const double* W = d.hpl + (size_t)e * 18;
        const double* xp = d.x + 6 * (size_t)d.e_pose[e];
#pragma unroll
        for (int c = 0; c < 3; c++) cl[c] -= W[c] * xp[0] + W[3 + c] * xp[1] + W[6 + c] * xp[2] + W[9 + c] * xp[3] + W[12 + c] * xp[4] + W[15 + c] * xp[5];
    }
    const double* Di = d.Dinv + 9 * (size_t)l;
    double* xl = d.x + d.sp + 3 * (size_t)l;
#pragma unroll
    for (int a = 0; a < 3; a++) xl[a] = Di[a * 3] * cl[0] + Di[a * 3 + 1] * cl[1] + Di[a * 3 + 2] * cl[2];
}
__global__ __launch_bounds__(256) void ba_backsub_kernel(CorbBADev d) { ba_backsub_body(d, blockIdx.x, threadIdx.x); }

// computeScale: sum_j x_j (lambda x_j + b_j)
__global__ __launch_bounds__(256) void ba_scale_kernel(CorbBADev d, double lambda, double* partial)
{
    __shared__ double red[4];
    double acc = 0;
    const int n = d.sp + 3 * d.nL;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += d.x[i] * (lambda * d.x[i] + d.b[i]);
    const double s = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// oplus: VertexSE3Expmap (T <- exp(dx) T, types_six_dof_expmap.h:73-76) and VertexSBAPointXYZ (X += dx)
__device__ __forceinline__ void ba_update_body(const CorbBADev& d, const int vbid, const int vtid)
{
    const int i = vbid * 256 + vtid;
    if (i < d.nP) {
        const int v = d.pose_vertex[i];
        double eq[4], et[3];
        se3_exp(d.x + 6 * (size_t)i, eq, et);
        se3_premul(eq, et, d.pose_q + 4 * (size_t)v, d.pose_t + 3 * (size_t)v);
    }
    if (i < d.nL) {
        const int v = d.point_vertex[i];
        d.pt[3 * (size_t)v] += d.x[d.sp + 3 * (size_t)i];
        d.pt[3 * (size_t)v + 1] += d.x[d.sp + 3 * (size_t)i + 1];
        d.pt[3 * (size_t)v + 2] += d.x[d.sp + 3 * (size_t)i + 2];
    }
}
__global__ __launch_bounds__(256) void ba_update_kernel(CorbBADev d) { ba_update_body(d, blockIdx.x, threadIdx.x); }
// The same with push() and computeScale folded in (one launch instead of copy + scale + reduction + update): the old estimate of every free vertex goes
// to the backup block (bak_off doubles further; fixed vertices never change, their backup is written once per call), the vertex's terms of
// sum_j x_j (lambda x_j + b_j) are summed per workgroup and finished by the last workgroup.
// backsub != 0 (lean form): the thread of landmark i first solves for its own increment (ba_backsub_lean_kernel's body: it needs the keyframes' increments only) --
// one launch less per LM trial.
__global__ __launch_bounds__(256) void ba_update_scale_kernel(CorbBADev d, double lambda, ptrdiff_t bak_off, double* partial, double* scale_out, int backsub)
{
    __shared__ double red[4];
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    if (d.ctl) lambda = d.ctl->lambda;
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    if (backsub && i < d.nL) ba_backsub_lean_one(d, i, false);
    if (i < d.nP) {
        const int v = d.pose_vertex[i];
        double* q = d.pose_q + 4 * (size_t)v; double* t = d.pose_t + 3 * (size_t)v;
#pragma unroll
        for (int a = 0; a < 4; a++) q[bak_off + a] = q[a];
#pragma unroll
        for (int a = 0; a < 3; a++) t[bak_off + a] = t[a];
#pragma unroll
        for (int a = 0; a < 6; a++) { const double x = d.x[6 * (size_t)i + a]; acc += x * (lambda * x + d.b[6 * (size_t)i + a]); }
    }
    if (i < d.nL) {
        const int v = d.point_vertex[i];
        double* X = d.pt + 3 * (size_t)v;
#pragma unroll
        for (int a = 0; a < 3; a++) { X[bak_off + a] = X[a]; const double x = d.x[d.sp + 3 * (size_t)i + a]; acc += x * (lambda * x + d.b[d.sp + 3 * (size_t)i + a]); }
    }
    ba_update_body(d, blockIdx.x, threadIdx.x);
    const double s = block_sum_256(acc, red);
    ba_finish_sum(s, partial, scale_out, d.red_tick, red);
}


// ------------------------------------------------------------------------------------------------
static inline int nblk(int n) { return (n + 255) / 256; }

void ba_launch_error(const CorbBADev& d, double* partial, int nparts, double* out, hipStream_t s, const int* ctl_bad, int ctl_epoch)
{
    hipLaunchKernelGGL(ba_error_kernel, dim3(nparts), dim3(256), 0, s, d, partial, out, ctl_bad, ctl_epoch);
}
int ba_build_lean_blocks(const CorbBADev& d) { const int lpb = d.nL <= 16384 ? 32 : 256; return (d.nL + lpb - 1) / lpb + (d.nE - d.nfree_edges + 255) / 256; }
// chi_partial (ba_build_lean_blocks() entries) / chi_out: lean form only -- the launch also evaluates and sums the edges' chi2 (see ba_build_lean_kernel)
void ba_launch_build(const CorbBADev& d, double* maxdiag_out, hipStream_t s, double* chi_partial, double* chi_out, const int* ctl_bad, int ctl_epoch)
{
    if (d.lean) { const int lpb = d.nL <= 16384 ? 32 : 256; const int nb = ba_build_lean_blocks(d); if (nb > 0) hipLaunchKernelGGL(ba_build_lean_kernel, dim3(nb), dim3(256), 0, s, d, lpb, chi_partial, chi_out, chi_partial ? ctl_bad : nullptr, ctl_epoch); }
    else {
    if (d.nE > 0) hipLaunchKernelGGL(ba_linearize_kernel, dim3(nblk(d.nE)), dim3(256), 0, s, d);
    if (d.nL > 0) hipLaunchKernelGGL(ba_sum_points_kernel, dim3(nblk(d.nL)), dim3(256), 0, s, d);
    }
    if (d.nP > 0 && d.hpp_scratch) hipLaunchKernelGGL(ba_hpp_scratch_kernel, dim3((d.nP + 3) / 4), dim3(256), 0, s, d);
    else if (d.nP > 0 && d.nP <= BA_SMALL_SPLIT_MAX_UNITS) hipLaunchKernelGGL(ba_hpp_mfma_kernel<BA_SMALL_SPLIT>, dim3(d.nP), dim3(64 * BA_SMALL_SPLIT), 0, s, d);
    else if (d.nP > 0) hipLaunchKernelGGL(ba_hpp_mfma_kernel<1>, dim3((d.nP + 3) / 4), dim3(256), 0, s, d);
    if (maxdiag_out) {
        (void)hipMemsetAsync(maxdiag_out, 0, sizeof(double), s);
        const int n = d.nP * 6 + d.nL * 3;
        hipLaunchKernelGGL(ba_maxdiag_kernel, dim3(std::max(1, std::min(1024, (n + 2047) / 2048))), dim3(256), 0, s, d, maxdiag_out);
    }
}
void ba_schur_mfma_launch(const CorbBADev& d, double lambda, int* bad, int epoch, hipStream_t s, int* with_rhs = nullptr);
// zero_S = 0: the caller knows that S still holds zeros outside the block pattern (pair-list kernels, which store every block of the pattern, and a solver
// that leaves S alone)
void ba_launch_schur(const CorbBADev& d, double lambda, int* bad, int epoch, int zero_S, hipStream_t s)
{
    if (zero_S) (void)hipMemsetAsync(d.S, 0, sizeof(double) * (size_t)d.sp * d.sp, s);
    // deterministic: every block of the pattern is written once by its wavefront (pair lists; a repeated (keyframe, map point) observation contributes all its cross products)
    if (d.nL > 0 && !d.lean) hipLaunchKernelGGL(ba_schur_prepare_kernel, dim3(nblk(d.nL)), dim3(256), 0, s, d, lambda, bad, epoch);
    int with_rhs = 0;
    if (d.nP > 0 || d.lean) ba_schur_mfma_launch(d, lambda, bad, epoch, s, &with_rhs);
    if (!with_rhs) ba_launch_reduced_rhs(d, s);
}
// bak != nullptr: state .. state + n_state (quaternions | translations | points, one block) is backed up to bak.  Up to BA_FUSED_UPDATE_BLOCKS workgroups
// the update kernel does it for the free vertices (the caller has copied the whole block once) together with computeScale; larger maps keep the
// separate copy / kernels (one ticket for thousands of workgroups would serialise them).
void ba_launch_backsub_update(const CorbBADev& d, double lambda, double* partial, int nparts, double* scale_out, double* state, double* bak, size_t n_state, hipStream_t s)
{
    const int nv = d.nP > d.nL ? d.nP : d.nL;
    const bool fused = bak && nv > 0 && nblk(nv) <= BA_FUSED_UPDATE_BLOCKS;
    if (d.nL > 0 && !(fused && d.lean)) { if (d.lean) hipLaunchKernelGGL(ba_backsub_lean_kernel, dim3(nblk(d.nL)), dim3(256), 0, s, d); else hipLaunchKernelGGL(ba_backsub_kernel, dim3(nblk(d.nL)), dim3(256), 0, s, d); }
    if (fused) {
        hipLaunchKernelGGL(ba_update_scale_kernel, dim3(nblk(nv)), dim3(256), 0, s, d, lambda, (ptrdiff_t)(bak - state), partial, scale_out, d.lean && d.nL > 0 ? 1 : 0);
        return;
    }
    if (bak) (void)hipMemcpyAsync(bak, state, n_state * 8, hipMemcpyDeviceToDevice, s);
    if (nparts == 1) hipLaunchKernelGGL(ba_scale_kernel, dim3(1), dim3(256), 0, s, d, lambda, scale_out);
    else {
        hipLaunchKernelGGL(ba_scale_kernel, dim3(nparts), dim3(256), 0, s, d, lambda, partial);
        hipLaunchKernelGGL(ba_reduce_kernel, dim3(1), dim3(256), 0, s, partial, nparts, scale_out);
    }
    const int n = d.nP > d.nL ? d.nP : d.nL;
    if (n > 0) hipLaunchKernelGGL(ba_update_kernel, dim3(nblk(n)), dim3(256), 0, s, d);
}

// ------------------------------------------------------------------------------------------------
// Small problems (local windows, small maps: sp <= BA_SMALL_SP, a few thousand observations): the WHOLE optimize() call -- every LM iteration,
// every trial, the dense Schur system (held in LDS, summed by block owners: no atomics), its Cholesky solve, the lambda control of optimization_algorithm_levenberg.cpp:61-164 --
// runs in ONE workgroup of 512 threads.  The multi-kernel form needs ~20 dependent stream operations and a host read-back per trial
// (0.25 ms); here a trial costs its barriers.  The phases are the bodies of the stand-alone kernels, executed by the 256-thread halves
// of the workgroup as virtual blocks.
#define SM_T 512                 // 8 waves: 256 VGPRs per thread (the edge Jacobians and the 27-term pose sums spill at 1024 threads / 128 VGPRs)
// value of lane `src` (wave-uniform index) through two v_readlane_b32: a few cycles, where a ds_bpermute shuffle costs an LDS round trip
__device__ __forceinline__ double small_readlane(double v, int src)
{ return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src)); }
__device__ __forceinline__ double small_block_sum(double v, double* red16)
{
    v = lx_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int w = 0; w < SM_T / 64; w++) s += red16[w];
    return s;
}
#define SMALL_RUN(G, CALL) do { for (int vb = q; vb < (G); vb += SM_T / 256) { CALL; } __syncthreads(); } while (0)
// Cholesky (left-looking: column k = dot products over the finished columns, all loads of a column independent) and the two triangular solves
// (row dot product + wave reduction, the solution kept in registers) of a dense system in LDS, in place, by ONE wavefront: its LDS operations
// execute in order, so the dependent steps need no workgroup barrier.  sp <= 128: a lane owns rows lane and lane + 64.  sm_S: row-major, the
// lower triangle is read; rhs: in = right-hand side, out = solution; *fail = 1 + the first pivot that is not positive (0 = success).
__device__ __forceinline__ void small_chol_solve_wave(double* sm_S, double* rhs, const int sp, const int lane, int* fail)
{
    const int r0 = lane, r1 = lane + 64;
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    double di0 = 0, di1 = 0;                                                   // 1 / L[r][r] of the lane's rows
    for (int k = 0; k < sp; k++) {
        const bool m0 = r0 >= k && r0 < sp, m1 = r1 >= k && r1 < sp;
        double s0 = m0 ? sm_S[r0 * sp + k] : 0.0, s1 = m1 ? sm_S[r1 * sp + k] : 0.0;
        const double* Lk = sm_S + k * sp;
        const double* L0 = sm_S + (m0 ? r0 : k) * sp; const double* L1 = sm_S + (m1 ? r1 : k) * sp;
        int c = 0;
        for (; c + 4 <= k; c += 4) {                                           // 12 independent loads, then the products
            const double a0 = Lk[c], a1 = Lk[c + 1], a2 = Lk[c + 2], a3 = Lk[c + 3];
            const double u0 = L0[c], u1 = L0[c + 1], u2 = L0[c + 2], u3 = L0[c + 3];
            const double v0 = L1[c], v1 = L1[c + 1], v2 = L1[c + 2], v3 = L1[c + 3];
            s0 -= u0 * a0 + u1 * a1 + u2 * a2 + u3 * a3; s1 -= v0 * a0 + v1 * a1 + v2 * a2 + v3 * a3;
        }
        for (; c < k; c++) { s0 -= L0[c] * Lk[c]; s1 -= L1[c] * Lk[c]; }
        const double piv = small_readlane(k < 64 ? s0 : s1, k & 63);                   // the diagonal element, held by the lane that owns row k
        double dk = 1.0;
        if (!(piv > 0)) { if (lane == 0 && !*fail) *fail = k + 1; } else dk = sqrt(piv);
        const double inv = 1.0 / dk;
        WAVE_SYNC();                                                           // every lane has read row k before column k is written
        if (m0) sm_S[r0 * sp + k] = (r0 == k) ? dk : s0 * inv;
        if (m1) sm_S[r1 * sp + k] = (r1 == k) ? dk : s1 * inv;
        if (r0 == k) di0 = inv;
        if (r1 == k) di1 = inv;
        WAVE_SYNC();
    }
    // L y = b, column form: once y[k] is known every lane adds its row's term; one shuffle per step
    double acc0 = 0, acc1 = 0;
    const double b0 = r0 < sp ? rhs[r0] : 0.0, b1 = r1 < sp ? rhs[r1] : 0.0;
    double y0 = 0, y1 = 0;
    for (int k = 0; k < sp; k++) {
        const double cand = k < 64 ? (b0 - acc0) * di0 : (b1 - acc1) * di1;
        const double yk = small_readlane(cand, k & 63);
        if (r0 == k) y0 = yk;
        if (r1 == k) y1 = yk;
        if (r0 > k && r0 < sp) acc0 += sm_S[r0 * sp + k] * yk;
        if (r1 > k && r1 < sp) acc1 += sm_S[r1 * sp + k] * yk;
    }
    // L' x = y, the same with row k of L
    acc0 = 0; acc1 = 0;
    double x0 = 0, x1 = 0;
    for (int k = sp - 1; k >= 0; k--) {
        const double cand = k < 64 ? (y0 - acc0) * di0 : (y1 - acc1) * di1;
        const double xk = small_readlane(cand, k & 63);
        if (r0 == k) x0 = xk;
        if (r1 == k) x1 = xk;
        if (r0 < k) acc0 += sm_S[k * sp + r0] * xk;
        if (r1 < k) acc1 += sm_S[k * sp + r1] * xk;
    }
    if (r0 < sp) rhs[r0] = x0;
    if (r1 < sp) rhs[r1] = x1;
#undef WAVE_SYNC
}

// Reduced system of a local window (sp <= 128, i.e. up to 21 free keyframes): S x = b by ONE workgroup in LDS instead of the rocSOLVER potrf /
// potrs kernel sequence, which at this size is ~150 us of launch and dependency latency per LM trial.  info = 1 + first non-positive pivot.
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: opt in once on every device that launches it
template <class K> static void ba_opt_in_lds(K kernel, int bytes, bool (&done)[64])
{
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || done[dev]) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[dev] = true;
}
__global__ __launch_bounds__(256) void ba_small_solve_kernel(CorbBADev d, int* info)
{
    extern __shared__ double small_solve_smem[];             // S[sp][sp] | rhs[sp]
    __shared__ int fail;
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    const int sp = d.sp, tid = threadIdx.x;
    double* sm_S = small_solve_smem; double* rhs = small_solve_smem + (size_t)sp * sp;
    if (tid == 0) fail = 0;
    for (int i = tid; i < sp * sp; i += 256) sm_S[i] = d.S[i];                  // symmetric, both triangles stored
    for (int i = tid; i < sp; i += 256) rhs[i] = d.x[i];
    __syncthreads();
    if (tid < 64) small_chol_solve_wave(sm_S, rhs, sp, tid, &fail);
    __syncthreads();
    for (int i = tid; i < sp; i += 256) d.x[i] = rhs[i];
    if (tid == 0) *info = fail;
}
// sp <= 32 (a local window with up to 5 free keyframes): the whole solve in the registers of ONE wavefront -- lane r holds row r of the lower triangle,
// column c of the factor needs row c, which v_readlane broadcasts; forward substitution column by column the same way, the transposed factor for the
// backward substitution through 8 KB of LDS.  No dependent LDS round trip per column: 29 -> ~10 us per solve on a 30 x 30 system.
__device__ __forceinline__ double small_readlane64(double v, int src)
{ return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src)); }
__global__ __launch_bounds__(64) void ba_small_solve32_kernel(CorbBADev d, int* info)
{
    __shared__ double tile[32][33];
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    const int sp = d.sp, lane = threadIdx.x, r = lane & 31;
    double L[32];
#pragma unroll
    for (int c = 0; c < 32; c++) L[c] = (r < sp && c < sp && c <= r) ? d.S[(size_t)r * sp + c] : (r == c ? 1.0 : 0.0);       // identity tail
    double bv = r < sp ? d.x[r] : 0.0;
    int bad = 0;
    double dinv[32];
#pragma unroll
    for (int c = 0; c < 32; c++) {
        double s0 = L[c], s1 = 0, s2 = 0, s3 = 0;            // four partial sums: shorter chains of dependent multiply-adds
#pragma unroll
        for (int m = 0; m < c; m++) {
            const double pr = L[m] * small_readlane64(L[m], c);
            if ((m & 3) == 0) s0 -= pr; else if ((m & 3) == 1) s1 -= pr; else if ((m & 3) == 2) s2 -= pr; else s3 -= pr;
        }
        s0 += (s1 + s2) + s3;
        const double piv = small_readlane64(s0, c);
        if (!(piv > 0.0) && !bad) bad = c + 1;
        const double lcc = sqrt(piv > 0.0 ? piv : 1.0);
        dinv[c] = 1.0 / lcc;
        L[c] = r == c ? lcc : (r > c ? s0 * dinv[c] : 0.0);
    }
    // L y = b, column form
    double yv = 0;
#pragma unroll
    for (int c = 0; c < 32; c++) {
        const double yc = small_readlane64(bv, c) * dinv[c];
        if (r == c) yv = yc;
        if (r > c) bv -= L[c] * yc;
    }
    // L' x = y: lane r needs column r of L
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < 32; c++) tile[r][c] = L[c];
    }
    __syncthreads();
    double LT[32];
#pragma unroll
    for (int c = 0; c < 32; c++) LT[c] = tile[c][r];
    double xv = 0, acc = yv;
#pragma unroll
    for (int c = 31; c >= 0; c--) {
        const double xc = small_readlane64(acc, c) * dinv[c];
        if (r == c) xv = xc;
        if (r < c) acc -= LT[c] * xc;
    }
    if (lane < sp) d.x[lane] = xv;
    if (lane == 0) *info = bad;
}
void ba_launch_small_solve(const CorbBADev& d, int* info, hipStream_t s)
{
    if (d.sp <= 32) { hipLaunchKernelGGL(ba_small_solve32_kernel, dim3(1), dim3(64), 0, s, d, info); return; }
    static bool attr_set[64] = {};
    ba_opt_in_lds(ba_small_solve_kernel, 140 * 1024, attr_set);
    hipLaunchKernelGGL(ba_small_solve_kernel, dim3(1), dim3(256), sizeof(double) * ((size_t)d.sp * d.sp + d.sp), s, d, info);
}

__global__ __launch_bounds__(SM_T) void ba_small_optimize_kernel(CorbBADev dg, CorbBASmall a)
{
    extern __shared__ double sm_S[];                    // sp x sp reduced system, then its Cholesky factor | sp right-hand side
    __shared__ double red16[16];
    __shared__ int flags[2];                             // Dinv not finite, pivot not positive
    const int tid = threadIdx.x, q = tid >> 8, t = tid & 255;
    CorbBADev d = dg; d.S = sm_S;                        // the bodies address S through a generic pointer: LDS here
    const int sp = d.sp, nE = d.nE, nL = d.nL, nP = d.nP;
    double* rhs = sm_S + (size_t)sp * sp;
    auto error_sum = [&]() -> double {                   // computeActiveErrors + activeRobustChi2 (ba_error_kernel)
        double acc = 0;
        for (int i = tid; i < nE; i += SM_T) {
            double err[3], Xc[3], rho[2];
            double c = edge_error(d, i, err, Xc);
            if (d.e_chi2) d.e_chi2[i] = c;
            if (d.robust) { huber(c, d.e_dim[i] == 2 ? d.delta2 : d.delta3, rho); c = rho[0]; }
            acc += c;
        }
        return small_block_sum(acc, red16);
    };
    double cur = error_sum();
    if (tid == 0) a.chi2_hist[0] = cur;
    double lambda = -1, ni = 2; int nBad = 0, it_done = 0, trials = 0; bool ok = true;
    for (int it = 0; it < a.iterations && ok && (nP + nL) > 0; it++) {
        // (no computeActiveErrors() here: the state is the one of the last error evaluation -- the initial one, or the accepted trial that ended the
        // previous iteration; an iteration that ends on a rejected trial terminates the loop -- so the per-edge chi2 are already those g2o would hold)
        double currentChi = cur;
        const double iniChi = currentChi; double tempChi = currentChi;
        SMALL_RUN((nE + 255) / 256, ba_linearize_body(d, vb, t));
        SMALL_RUN((nL + 255) / 256, ba_sum_points_body(d, vb, t));
        SMALL_RUN((nP + 3) / 4, ba_sum_poses_body(d, vb, t));
        if (it == 0) {                                   // computeLambdaInit, _tau = 1e-5
            double m = 0;
            for (int i = tid; i < nP * 6; i += SM_T) m = fmax(m, fabs(d.Hpp[36 * (size_t)(i / 6) + 7 * (i % 6)]));
            for (int i = tid; i < nL * 3; i += SM_T) m = fmax(m, fabs(d.Hll[9 * (size_t)(i / 3) + 4 * (i % 3)]));
            m = lx_wave_max(m);
            __syncthreads();
            if ((tid & 63) == 0) red16[tid >> 6] = m;
            __syncthreads();
            double mm = 0;
            for (int w = 0; w < SM_T / 64; w++) mm = fmax(mm, red16[w]);
            lambda = 1e-5 * mm; ni = 2; nBad = 0;
        }
        double rho = 0; int qmax = 0;
        do {
            __syncthreads();                             // every thread has read the previous trial's flags
            for (size_t i = tid; i < a.n_state; i += SM_T) a.state_bak[i] = a.state[i];          // push()
            for (int i = tid; i < sp * sp; i += SM_T) sm_S[i] = 0.0;
            if (tid < 2) flags[tid] = 0;
            __syncthreads();
            SMALL_RUN((nP * 36 + 255) / 256, ba_s_diag_body(d, vb, t, lambda));
            SMALL_RUN((nL + 255) / 256, ba_schur_prepare_body(d, vb, t, lambda, &flags[0], 1));
            // Schur products straight into the LDS system, OWNER form (no atomics: fixed summation order, bit-identical runs): a work item is one 3 x 3 quarter of
            // a block (pa, pb <= pa) of the lower triangle -- the triangle the Cholesky below reads; it walks keyframe pa's edges in list order and, per edge,
            // the free-pose edges of that edge's landmark for keyframe pb:  S(pa, pb) -= (W_a Dinv) W_b'.  (Round 2 scattered every product with ds_add_f64.)
            for (int w = tid; w < 2 * nP * (nP + 1); w += SM_T) {
                const int bi = w >> 2, qd = w & 3;
                int pa = 0; while ((pa + 1) * (pa + 2) / 2 <= bi) pa++;
                const int pb = bi - pa * (pa + 1) / 2, r0 = 3 * (qd >> 1), c0 = 3 * (qd & 1);
                double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int ii = d.poff[pa]; ii < d.poff[pa + 1]; ii++) {
                    const int ea = d.pedge[ii], l = d.e_point[ea];
                    if (l < 0) continue;
                    const int e0 = d.loff[l], nf = d.lnfree[l];
                    for (int j = 0; j < nf; j++) {
                        if (d.e_pose[e0 + j] != pb) continue;
                        const double* Di = d.Dinv + 9 * (size_t)l;
                        const double* Wa = d.hpl + (size_t)ea * 18 + r0 * 3; const double* Wb = d.hpl + (size_t)(e0 + j) * 18 + c0 * 3;
#pragma unroll
                        for (int r = 0; r < 3; r++) {
                            const double bd0 = Wa[r * 3] * Di[0] + Wa[r * 3 + 1] * Di[3] + Wa[r * 3 + 2] * Di[6];
                            const double bd1 = Wa[r * 3] * Di[1] + Wa[r * 3 + 1] * Di[4] + Wa[r * 3 + 2] * Di[7];
                            const double bd2 = Wa[r * 3] * Di[2] + Wa[r * 3 + 1] * Di[5] + Wa[r * 3 + 2] * Di[8];
#pragma unroll
                            for (int c = 0; c < 3; c++) acc[r * 3 + c] += bd0 * Wb[c * 3] + bd1 * Wb[c * 3 + 1] + bd2 * Wb[c * 3 + 2];
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) sm_S[(6 * pa + r0 + r) * sp + 6 * pb + c0 + c] -= acc[r * 3 + c];
            }
            __syncthreads();
            SMALL_RUN((nP + 3) / 4, ba_reduced_rhs_body(d, vb, t));
            // Cholesky (left-looking: column k = dot products over the finished columns, all loads of a column independent) and the two triangular
            // solves (row dot product + wave reduction, the solution kept in registers), in place in LDS, by ONE wavefront: its LDS operations execute
            // in order, so the dependent steps need no workgroup barrier.  sp <= 128: a lane owns rows lane and lane + 64.
            for (int i = tid; i < sp; i += SM_T) rhs[i] = d.x[i];
            __syncthreads();
            if (tid < 64) small_chol_solve_wave(sm_S, rhs, sp, tid, &flags[1]);
            __syncthreads();
            for (int i = tid; i < sp; i += SM_T) d.x[i] = rhs[i];
            __syncthreads();
            SMALL_RUN((nL + 255) / 256, ba_backsub_body(d, vb, t));
            double sacc = 0;
            for (int i = tid; i < sp + 3 * nL; i += SM_T) sacc += d.x[i] * (lambda * d.x[i] + d.b[i]);      // computeScale
            double scale = small_block_sum(sacc, red16);
            SMALL_RUN(((nP > nL ? nP : nL) + 255) / 256, ba_update_body(d, vb, t));
            const double newChi = error_sum();
            const bool ok2 = flags[0] == 0 && flags[1] == 0;
            tempChi = ok2 ? newChi : DBL_MAX;
            if (!ok2) scale = 0;
            rho = currentChi - tempChi;
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3.0);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha); ni = 2; currentChi = tempChi; cur = tempChi;      // discardTop()
            } else {
                lambda *= ni; ni *= 2;                                                                 // pop()
                __syncthreads();
                for (size_t i = tid; i < a.n_state; i += SM_T) a.state[i] = a.state_bak[i];
                __syncthreads();
                if (!ok2) (void)error_sum();             // failed solve: g2o evaluated the errors at the unchanged state
            }
            qmax++; trials++;
        } while (rho < 0 && qmax < 10);
        it_done++;
        if (tid == 0) { a.chi2_hist[it_done] = currentChi; a.lambda_hist[it_done - 1] = lambda; }
        if (qmax == 10 || rho == 0) { ok = false; continue; }                                          // Terminate
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;                               // ORB-SLAM2 stop rule (:155-161)
        if (nBad >= 3) ok = false;
    }
    if (tid == 0) { a.counters[0] = it_done; a.counters[1] = trials; }
}
// One LM trial's decision on the device (BALMCtl): what the host loop of ba_lm_device does with the trial's read-back when the trial is ACCEPTED
// (optimization_algorithm_levenberg.cpp:120-141: rho, lambda *= max(1/3, 1 - (2 rho - 1)^3), ni = 2; then ORB-SLAM2's stop rule); anything else stops the chain.
__global__ void ba_lm_ctl_kernel(CorbBADev d, const double* scal, const int* bad, int epoch)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ba_lm_decide(d.ctl, scal, bad, epoch);
}
void ba_launch_lm_ctl(const CorbBADev& d, const double* scal, const int* bad, int epoch, hipStream_t s)
{
    hipLaunchKernelGGL(ba_lm_ctl_kernel, dim3(1), dim3(64), 0, s, d, scal, bad, epoch);
}
// The start of an optimize() call inside a chain (BALMCtl::begin): computeLambdaInit (tau = 1e-5 times the largest diagonal entry, optimization_algorithm_levenberg.cpp:166-178)
// and the chi2 of the start estimates, as the host loop of ba_lm_device takes them from its two read-backs.
__global__ void ba_lm_begin_kernel(CorbBADev d, const double* scal)
{
    BALMCtl* c = d.ctl;
    if (threadIdx.x != 0 || blockIdx.x != 0 || c->stop) return;
    c->lambda = 1e-5 * scal[1]; c->ni = 2; c->nBad = 0;
    c->currentChi = c->chi0 = scal[0];
}
void ba_launch_lm_begin(const CorbBADev& d, const double* scal, hipStream_t s)
{
    hipLaunchKernelGGL(ba_lm_begin_kernel, dim3(1), dim3(64), 0, s, d, scal);
}
// The classification after an optimize() call of a staged solve (corb_ba_solve_staged's loop over the edges, BAStageDev): one thread per edge of the session's graph.
__global__ __launch_bounds__(256) void ba_stage_classify_kernel(CorbBADev d, BAStageDev a)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d.nE) return;
    const bool on = a.act_in == nullptr || a.act_in[j] != 0;   // (NULL: the first classification, every edge was active)
    double last = on ? d.e_chi2[j] : a.last[j];              // (an edge that is switched off has no computeError(): it keeps the chi2 it had when it last was active)
    double depth = 1.0;
    if (a.check_depth || a.recompute_inactive) {
        double err[3], Xc[3];
        const double fresh = edge_error(d, j, err, Xc);
        depth = Xc[2];
        if (!on && a.recompute_inactive) last = fresh;
    }
    bool now = on;
    if (on || a.allow_reactivate) {
        const bool mono = d.e_dim[j] == 2;
        bool out = a.float_compare ? ((float)last > (mono ? a.th_mono : a.th_stereo)) : (last > (mono ? a.thd_mono : a.thd_stereo));
        if (a.check_depth && !(depth > 0.0)) out = true;
        now = !out;
    }
    a.last[j] = last; a.act_out[j] = now ? 1 : 0; a.e_w[j] = now ? a.w0[j] : 0.0;
}
void ba_launch_stage_classify(const CorbBADev& d, const BAStageDev& a, hipStream_t s)
{
    if (d.nE > 0) hipLaunchKernelGGL(ba_stage_classify_kernel, dim3(nblk(d.nE)), dim3(256), 0, s, d, a);
}
void ba_launch_small_optimize(const CorbBADev& d, const CorbBASmall& a, hipStream_t s)
{
    static bool attr_set[64] = {};
    ba_opt_in_lds(ba_small_optimize_kernel, 150 * 1024, attr_set);
    hipLaunchKernelGGL(ba_small_optimize_kernel, dim3(1), dim3(SM_T), sizeof(double) * ((size_t)d.sp * d.sp + d.sp + 1), s, d, a);
}

// ------------------------------------------------------------------------------------------------
// Block-sparse reduced camera system + preconditioned conjugate gradients (solver 2).
// S (BSR, 6x6 blocks) = blockdiag(Hpp + lambda I) - sum_l W_l Dinv_l W_l' ; M = blockdiag(S)  (block Jacobi).
// Two kernels per CG iteration, all scalars (alpha, beta, residual) stay on the device:
//   pcg_spmv : beta = rz_new/rz_old ; p = z + beta p_old (double-buffered) ; q = S p ; partial p.q
//   pcg_step : alpha = rz/(p.q) ; x += alpha p ; r -= alpha q ; z = Minv r ; partial r.z, r.r ; convergence flag
// Minv = inverse of each 6x6 diagonal block (Cholesky L L', then inverse via two triangular solves)
__global__ __launch_bounds__(256) void ba_minv_kernel(CorbBADev d)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d.nP) return;
    double A[36], Li[36];
    const double* src = d.bsr_val + (size_t)d.bsr_diag[k] * 36;
#pragma unroll
    for (int i = 0; i < 36; i++) A[i] = src[i];
    bool ok = true;
    for (int j = 0; j < 6; j++) {                       // Cholesky, lower in A
        double dj = A[j * 6 + j];
        for (int t = 0; t < j; t++) dj -= A[j * 6 + t] * A[j * 6 + t];
        if (!(dj > 0)) { ok = false; dj = 1; }
        dj = sqrt(dj); A[j * 6 + j] = dj;
        for (int i = j + 1; i < 6; i++) { double v = A[i * 6 + j]; for (int t = 0; t < j; t++) v -= A[i * 6 + t] * A[j * 6 + t]; A[i * 6 + j] = v / dj; }
    }
    for (int c = 0; c < 6; c++)                          // Li = L^-1 (lower)
        for (int i = 0; i < 6; i++) {
            double v = (i == c) ? 1.0 : 0.0;
            for (int t = c; t < i; t++) v -= A[i * 6 + t] * Li[t * 6 + c];
            Li[i * 6 + c] = i < c ? 0.0 : v / A[i * 6 + i];
        }
    double* o = d.Minv + (size_t)k * 36;
    for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) { double v = 0; for (int t = (a > c ? a : c); t < 6; t++) v += Li[t * 6 + a] * Li[t * 6 + c]; o[a * 6 + c] = v; }
    if (!ok) d.cg_flag[1] = 1;
}

// three block sums with one barrier pair
__device__ __forceinline__ void block_sum3_256(double& a, double& b, double& c, double* red12)
{
    a = lx_wave_sum(a); b = lx_wave_sum(b); c = lx_wave_sum(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red12[w] = a; red12[4 + w] = b; red12[8 + w] = c; }
    __syncthreads();
    a = red12[0] + red12[1] + red12[2] + red12[3]; b = red12[4] + red12[5] + red12[6] + red12[7]; c = red12[8] + red12[9] + red12[10] + red12[11];
}

// deterministic reduction of n per-workgroup partials (every workgroup does it in the same order)
__device__ __forceinline__ double cg_reduce_parts(const double* part, int n, double* red)
{
    double v = 0;
    for (int t = threadIdx.x; t < n; t += 256) v += part[t];
    return block_sum_256(v, red);
}

// Large systems (cg_two_level): the partials of every 64 consecutive workgroups are summed (wave tree: fixed order, deterministic) by the LAST of those
// workgroups to publish its own, into a second-level array a hundred times shorter that every consumer workgroup can afford to sum itself -- no
// one-workgroup reduction launch between the two kernels of a CG iteration (at 25 000 keyframes those launches were 15 of the 83 us of kernel time per
// iteration).  No agent-scope fence: on this multi-XCD part a release fence writes back the XCD's whole L2, and 12 500 workgroups doing so made a CG
// kernel 20x slower.  Only the partials cross workgroups inside a kernel, and they travel through agent-scope atomics (performed at the device's
// coherence point): thread 0 publishes, waits for the acknowledgement, takes the group's ticket; the last taker's loads are agent-scope atomics issued
// after the ticket's value has come back.  One ticket per GROUP: a single ticket for all workgroups serialised 12 500 same-address atomics (+49 us
// per kernel at 50 000 keyframes).  The ticket is reset by its last taker.
// The stop tolerance (|r|^2 <= tol^2 |b|^2) of the solve in progress lives on the device (set by ba_pcg_zero_x_kernel): the captured chunks of CG iterations
// serve every tolerance, and the LM loop may change it from one trial to the next (corb_ba.cpp: the default policy tightens it after a rejected trial).
#define CG_TOL2(d) ((d).cg_scal[5])
#define CG_GROUP 64
#define CG_TICK_STRIDE 64      // ints between two tickets: one ticket per 256 bytes, so that the groups' atomics go to different L2 channels
#define CG2_RZ(d, par) ((d).cg_part2 + (size_t)(par) * (d).cg_ngrp)
#define CG2_RR(d, par) ((d).cg_part2 + (size_t)(2 + (par)) * (d).cg_ngrp)
#define CG2_PQ(d) ((d).cg_part2 + (size_t)4 * (d).cg_ngrp)         /* cg_ngrp_spmv entries */
__device__ __forceinline__ void cg_publish(double* slot, double v) { __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Round 3: a THIRD level.  The consumers used to sum the second-level array themselves (a hundred to two hundred loads, an LDS exchange and two
// workgroup barriers at the top of every workgroup of the next kernel), and all four wavefronts of a workgroup sat through thread 0's two device-scope
// round trips at its end; tools/ubench/stream_patterns.hip streams the same bytes with the same dependent index / gather chain at 5.7 TB/s against the
// 4.3 TB/s of the real SpMV, i.e. a third of the kernel was its head and tail.  Now wavefronts 1-3 leave after the workgroup sum, wavefront 0 alone takes
// the group ticket, the group's last wavefront publishes the group sums and takes ONE more ticket, and the last of those sums the groups (fixed order)
// into cg_fin: the next kernel reads three scalars.  Same guarantees: agent-scope atomics carry the partials, plain loads only across kernel boundaries.
#define CG_FIN_RZ(d, par) ((d).cg_fin + (par))
#define CG_FIN_RR(d, par) ((d).cg_fin + 2 + (par))
#define CG_FIN_PQ(d) ((d).cg_fin + 4)
#define CG_TICK3(d, spmv) ((d).cg_tick + ((size_t)(d).cg_ngrp + (d).cg_ngrp_spmv + ((spmv) ? 1 : 0)) * CG_TICK_STRIDE)
// called by ONE whole wavefront of every workgroup after its lane 0 has published the workgroup's partial(s) in part0 (and part1; nullptr = none).  g0a / g0b
// (g1a / g1b): second-level arrays that receive the group sums of part0 (part1), f0a / f0b (f1a / f1b): the final scalars; the b pointers may be null.
// nwg: the workgroups that take part (0 = the whole grid; a launch that carries other work behind them names their number)
__device__ __forceinline__ void cg_tree_reduce(int* tick, int* tick3, int ngrp, const double* part0, const double* part1,
                                               double* g0a, double* g0b, double* g1a, double* g1b, double* f0a, double* f0b, double* f1a, double* f1b, int nwg = 0)
{
    const int lane = threadIdx.x & 63, grp = blockIdx.x / CG_GROUP;
    const int first = grp * CG_GROUP, n_in = min(CG_GROUP, (nwg ? nwg : (int)gridDim.x) - first);
    int last = 0;
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the fence above orders the compiler; this waits for the stores' acknowledgements)
        last = __hip_atomic_fetch_add(tick + (size_t)grp * CG_TICK_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_in - 1;
    }
    if (!__shfl(last, 0)) return;
    double v0 = lane < n_in ? __hip_atomic_load(part0 + first + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    double v1 = (part1 && lane < n_in) ? __hip_atomic_load(part1 + first + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    v0 = lx_wave_sum(v0); v1 = lx_wave_sum(v1);
    int last3 = 0;
    if (lane == 0) {
        tick[(size_t)grp * CG_TICK_STRIDE] = 0;
        cg_publish(g0a + grp, v0); if (g0b) cg_publish(g0b + grp, v0);
        if (part1) { cg_publish(g1a + grp, v1); if (g1b) cg_publish(g1b + grp, v1); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last3 = __hip_atomic_fetch_add(tick3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp - 1;
    }
    if (!__shfl(last3, 0)) return;
    double w0 = 0, w1 = 0;
    for (int t = lane; t < ngrp; t += 64) {
        w0 += __hip_atomic_load(g0a + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (part1) w1 += __hip_atomic_load(g1a + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    w0 = lx_wave_sum(w0); w1 = lx_wave_sum(w1);
    if (lane == 0) {
        *tick3 = 0;
        *f0a = w0; if (f0b) *f0b = w0;
        if (part1) { *f1a = w1; if (f1b) *f1b = w1; }
    }
}

// Workgroup sum of two values WITHOUT a barrier: a wavefront leaves its own sums in LDS and takes an LDS ticket; the last of the four gets the workgroup's
// sums (added in wavefront order: deterministic) and goes on to publish them, the others are done.  With __syncthreads() here every wavefront of a
// workgroup held its slot until the slowest of the four had its last operands (tools/ubench/stream_patterns.hip: 76 -> 91 us for the SpMV's byte stream).
// cnt must have been zeroed before any wavefront can get here (top of the kernel + one barrier, where all four still run together).
__device__ __forceinline__ bool cg_wave_handoff(double& v0, double& v1, double* red8, int* cnt)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v0 = lx_wave_sum(v0); v1 = lx_wave_sum(v1);
    int t = 0;
    if (lane == 0) {
        red8[w] = v0; red8[4 + w] = v1;
        t = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (__shfl(t, 0) != 3) return false;
    v0 = red8[0] + red8[1] + red8[2] + red8[3]; v1 = red8[4] + red8[5] + red8[6] + red8[7];
    return true;
}

// Partial-sum layout (np = cg_nparts workgroups): pq[np] | rz[2][np] | rr[2][np].  The r.z / r.r partials are
// double-buffered by iteration parity, so every workgroup of every kernel can recompute alpha, beta and the
// convergence test from the same numbers in the same order -- no scalar hand-off kernel, no flags to read.
#define CG_RZ(d, par) ((d).cg_part + (size_t)(par) * (d).cg_nparts)
#define CG_RR(d, par) ((d).cg_part + (size_t)(2 + (par)) * (d).cg_nparts)
#define CG_PQ(d) ((d).cg_part + (size_t)4 * (d).cg_nparts)          /* cg_nparts_spmv entries */

// r0 = b_schur (held in x), z0 = Minv r0, partial r.z and r.r into parity-1 slots ("iteration -1")
__global__ __launch_bounds__(256) void ba_pcg_init_kernel(CorbBADev d)
{
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;        // scalar row
    double rz = 0, rr = 0;
    if (i < d.sp) {
        const int k = i / 6, a = i % 6;
        const double* r = d.x + 6 * (size_t)k;           // b_schur
        const double* Mi = d.Minv + (size_t)k * 36 + a * 6;
        const double z = Mi[0] * r[0] + Mi[1] * r[1] + Mi[2] * r[2] + Mi[3] * r[3] + Mi[4] * r[4] + Mi[5] * r[5];
        d.cg_r[0][i] = r[a]; d.cg_z[i] = z; d.cg_p[1][i] = 0.0; d.cg_q[i] = 0.0;
        rz = r[a] * z; rr = r[a] * r[a];
    }
    const double s1 = block_sum_256(rz, red);
    const double s2 = block_sum_256(rr, red);
    if (threadIdx.x == 0) { cg_publish(&CG_RZ(d, 1)[blockIdx.x], s1); cg_publish(&CG_RR(d, 1)[blockIdx.x], s2); cg_publish(&CG_RZ(d, 0)[blockIdx.x], s1); cg_publish(&CG_RR(d, 0)[blockIdx.x], s2); }
    if (d.cg_two_level && threadIdx.x < 64)
        cg_tree_reduce(d.cg_tick, CG_TICK3(d, 0), d.cg_ngrp, CG_RZ(d, 1), CG_RR(d, 1), CG2_RZ(d, 0), CG2_RZ(d, 1), CG2_RR(d, 0), CG2_RR(d, 1),
                       CG_FIN_RZ(d, 0), CG_FIN_RZ(d, 1), CG_FIN_RR(d, 0), CG_FIN_RR(d, 1));
}

// ---- block-Jacobi with large blocks (pc_g poses per block): the inverse blocks come from ba_pc_invert_kernel ----
// z = Dinv_b r for the BA_PC_ROWS rows of this workgroup (block b = blockIdx.x / split, slice blockIdx.x % split); rn = the block's
// residual in LDS.  Adds this thread's share of r.z and r.r (summed over the workgroup by the caller).
template <class T> __device__ __forceinline__ void pc_apply_rows_t(const CorbBADev& d, const T* pc, const double* rn, int b, int slice, double& rz, double& rr)
{
    // 16 lanes per row (consecutive lanes read consecutive elements: 128- / 64-byte runs), 4 rows per wave and pass, 3 passes for the 48 rows;
    // every product of a lane is an independent load, the 16 partial sums meet in 4 exchanges (fixed order: deterministic).  The inverse blocks are
    // the largest array a CG iteration reads (230 MB of 670 at 50 000 keyframes in double precision); a preconditioner rounded to single precision
    // is as good a preconditioner (products and sums stay double)
    const int n = d.pc_gb, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l16 = lane & 15, rsub = lane >> 4, row0 = b * n;
    const T* D = pc + (size_t)b * n * n;
#pragma unroll
    for (int pass = 0; pass < BA_PC_ROWS / 16; pass++) {
        const int t = slice * BA_PC_ROWS + pass * 16 + wave * 4 + rsub;
        const T* Dr = D + (size_t)t * n;                 // symmetric: row t == column t
        double acc = 0;
#pragma unroll 6
        for (int c = l16; c < n; c += 16) acc += (double)Dr[c] * rn[c];
        acc += lx_xor<1>(acc); acc += lx_xor<2>(acc); acc += lx_xor<4>(acc); acc += lx_xor<8>(acc);
        if (l16 == 0 && row0 + t < d.sp) { d.cg_z[row0 + t] = acc; rz += rn[t] * acc; rr += rn[t] * rn[t]; }
    }
}
__device__ __forceinline__ void pc_apply_rows(const CorbBADev& d, const double* rn, int b, int slice, double& rz, double& rr)
{
    if (d.pc_inv32) pc_apply_rows_t<float>(d, d.pc_inv32, rn, b, slice, rz, rr);
    else pc_apply_rows_t<double>(d, d.pc_inv, rn, b, slice, rz, rr);
}
__global__ __launch_bounds__(256) void ba_pcg_init_big_kernel(CorbBADev d)
{
    __shared__ double red[12];
    extern __shared__ double pc_rn[];
    const int split = d.pc_split, b = blockIdx.x / split, slice = blockIdx.x - b * split, row0 = b * d.pc_gb;
    for (int t = threadIdx.x; t < d.pc_gb; t += 256) {
        const int i = row0 + t;
        const double v = i < d.sp ? d.x[i] : 0.0;         // b_schur
        pc_rn[t] = v;
        if (i < d.sp && (split == 1 || t / BA_PC_ROWS == slice)) { d.cg_r[0][i] = v; d.cg_p[1][i] = 0.0; d.cg_q[i] = 0.0; }
    }
    __syncthreads();
    double rz = 0, rr = 0, dummy = 0;
    if (split == 1) { for (int sl = 0; sl < d.pc_gb / BA_PC_ROWS; sl++) pc_apply_rows(d, pc_rn, b, sl, rz, rr); }      // (once per solve: the square form)
    else pc_apply_rows(d, pc_rn, b, slice, rz, rr);
    block_sum3_256(rz, rr, dummy, red);
    if (threadIdx.x == 0) { cg_publish(&CG_RZ(d, 1)[blockIdx.x], rz); cg_publish(&CG_RR(d, 1)[blockIdx.x], rr); cg_publish(&CG_RZ(d, 0)[blockIdx.x], rz); cg_publish(&CG_RR(d, 0)[blockIdx.x], rr); }
    if (d.cg_two_level && threadIdx.x < 64)
        cg_tree_reduce(d.cg_tick, CG_TICK3(d, 0), d.cg_ngrp, CG_RZ(d, 1), CG_RR(d, 1), CG2_RZ(d, 0), CG2_RZ(d, 1), CG2_RR(d, 0), CG2_RR(d, 1),
                       CG_FIN_RZ(d, 0), CG_FIN_RZ(d, 1), CG_FIN_RR(d, 0), CG_FIN_RR(d, 1));
}
// rows / inverse-block operands of one lane of pc_apply_rows_t, loaded ahead of everything that depends on the scalars of the iteration
template <class T> struct PcOperands { T v[BA_PC_ROWS / 16][6]; };
template <class T> __device__ __forceinline__ void pc_load_rows(const CorbBADev& d, const T* pc, int b, int slice, PcOperands<T>& o)
{
    const int n = d.pc_gb, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l16 = lane & 15, rsub = lane >> 4;
    const T* D = pc + (size_t)b * n * n;
#pragma unroll
    for (int pass = 0; pass < BA_PC_ROWS / 16; pass++) {
        const T* Dr = D + (size_t)(slice * BA_PC_ROWS + pass * 16 + wave * 4 + rsub) * n;
#pragma unroll
        for (int u = 0; u < 6; u++) o.v[pass][u] = Dr[l16 + 16 * u < n ? l16 + 16 * u : l16];     // (unconditional loads: a load under a condition becomes a branch and a wait each)
    }
}
template <class T> __device__ __forceinline__ void pc_apply_loaded(const CorbBADev& d, const PcOperands<T>& o, const double* rn, int b, int slice, double& rz, double& rr)
{
    const int n = d.pc_gb, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l16 = lane & 15, rsub = lane >> 4, row0 = b * n;
#pragma unroll
    for (int pass = 0; pass < BA_PC_ROWS / 16; pass++) {
        const int t = slice * BA_PC_ROWS + pass * 16 + wave * 4 + rsub;
        double acc = 0;
#pragma unroll
        for (int u = 0; u < 6; u++) if (l16 + 16 * u < n) acc += (double)o.v[pass][u] * rn[l16 + 16 * u];
        acc += lx_xor<1>(acc); acc += lx_xor<2>(acc); acc += lx_xor<4>(acc); acc += lx_xor<8>(acc);
        if (l16 == 0 && row0 + t < d.sp) { d.cg_z[row0 + t] = acc; rz += rn[t] * acc; rr += rn[t] * rn[t]; }
    }
}
template <class T> __device__ __forceinline__ void ba_pcg_step_big_body(const CorbBADev& d, const T* pc, int par, double* pc_rn, double* red, int* cnt)
{
    // No workgroup barrier after the first instructions: the inverse-block operands are in flight before the scalars of the iteration are known, every
    // wavefront keeps its OWN copy of the block's new residual in LDS (96 values: recomputing them four times costs less than waiting for the other three
    // wavefronts), and the workgroup's sums pass through cg_wave_handoff.  (pc_gb <= 96: six 16-lane strides per row.)
    const int split = d.pc_gb / BA_PC_ROWS, b = blockIdx.x / split, slice = blockIdx.x - b * split, row0 = b * d.pc_gb;
    if (threadIdx.x == 0) *cnt = 0;
    __syncthreads();                                      // (before the first load: a barrier also waits for the wavefront's outstanding loads)
    PcOperands<T> ops;
    pc_load_rows(d, pc, b, slice, ops);
    if (d.cg_flag[1] || d.cg_flag[0]) return;             // failed, or converged in an EARLIER kernel (the r.r slot of the other parity is stale then)
    double rr_prev = 0, pq = 0, rz = 0;
    if (d.cg_two_level) { rr_prev = *CG_FIN_RR(d, par ^ 1); rz = *CG_FIN_RZ(d, par ^ 1); pq = *CG_FIN_PQ(d); }
    else {
        for (int t = threadIdx.x; t < d.cg_nparts; t += 256) { rr_prev += CG_RR(d, par ^ 1)[t]; rz += CG_RZ(d, par ^ 1)[t]; }
        for (int t = threadIdx.x; t < d.cg_nparts_spmv; t += 256) pq += CG_PQ(d)[t];
        block_sum3_256(rr_prev, pq, rz, red + 8);
    }
    if (rr_prev <= CG_TOL2(d) * d.cg_scal[2]) return;                               // converged: spmv of this iteration did not run
    if (!(pq > 0)) { if (blockIdx.x == 0 && threadIdx.x == 0) d.cg_flag[1] = 1; return; }    // not positive definite
    const double alpha = rz / pq;
    const double* p = d.cg_p[par];
    const double* rold = d.cg_r[par]; double* rnew = d.cg_r[par ^ 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* my_rn = pc_rn + (size_t)wave * d.pc_gb;
    for (int t = lane; t < d.pc_gb; t += 64) {                                // the whole block's new residual, once per wavefront
        const int i = row0 + t;
        const double v = i < d.sp ? rold[i] - alpha * d.cg_q[i] : 0.0;
        my_rn[t] = v;
        if (wave == 0 && i < d.sp && t / BA_PC_ROWS == slice) { d.x[i] += alpha * p[i]; rnew[i] = v; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double rzn = 0, rrn = 0;
    pc_apply_loaded(d, ops, my_rn, b, slice, rzn, rrn);
    if (!cg_wave_handoff(rzn, rrn, red, cnt)) return;
    if ((threadIdx.x & 63) == 0) { cg_publish(&CG_RZ(d, par)[blockIdx.x], rzn); cg_publish(&CG_RR(d, par)[blockIdx.x], rrn); }
    if (d.cg_two_level)
        cg_tree_reduce(d.cg_tick, CG_TICK3(d, 0), d.cg_ngrp, CG_RZ(d, par), CG_RR(d, par), CG2_RZ(d, par), nullptr, CG2_RR(d, par), nullptr,
                       CG_FIN_RZ(d, par), nullptr, CG_FIN_RR(d, par), nullptr, d.cg_nparts);
}
// The same step on the blocks' upper triangles (pc_pack32; round 5): ONE workgroup per block, a wavefront per tile (ti = wave, wave + 4, ...).  A 16 x 16 tile (I, J) is one
// coalesced 1 KB load -- lane l holds row l % 16, columns 4 (l / 16) .. + 3 (round 6; round 5: row l / 4, columns 4 (l % 4)) --; its row sums (two exchanges over the 4
// column groups) go to z_I, and for I != J its column sums (four exchanges over the 16 rows, inside a DPP row) to z_J: the transposed tile is never read.  A wavefront adds into its OWN copy of z in LDS in program order, the last wavefront
// through the ticket adds the four copies in wavefront order (deterministic) and forms r.z / r.r.  96 x 96 blocks: 21 of 36 tiles = 21.5 of 36.9 KB per block and iteration.
template <int NT> __device__ __forceinline__ void ba_pcg_step_sym_body(const CorbBADev& d, int par, double* pc_rn, double* yw, double* red, int* cnt)
{
    constexpr int N = 16 * NT, T = NT * (NT + 1) / 2, TPW = (T + 3) / 4;
    const int b = blockIdx.x, row0 = b * N, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) *cnt = 0;
    __syncthreads();                                      // (before the first load: a barrier also waits for the wavefront's outstanding loads)
    const float4* P = reinterpret_cast<const float4*>(d.pc_pack32 + (size_t)b * T * 256);
    float4 m[TPW];
#pragma unroll
    for (int u = 0; u < TPW; u++) { const int ti = wave + 4 * u; m[u] = P[(size_t)(ti < T ? ti : 0) * 64 + lane]; }      // in flight before the scalars of the iteration are known
    // ... and so are the block's slices of r, q (and p for the wavefront that updates x): their addresses do not depend on alpha
    constexpr int NV = (N + 63) / 64;
    double v_r[NV], v_q[NV], v_p[NV];
    {
        const double* rold0 = d.cg_r[par]; const double* p0 = d.cg_p[par];
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const int t = lane + 64 * u, i = row0 + t, ii = (t < N && i < d.sp) ? i : 0;
            v_r[u] = rold0[ii]; v_q[u] = d.cg_q[ii]; v_p[u] = wave == 0 ? p0[ii] : 0.0;
        }
    }
    // the flags and the iteration's scalars as ONE batch of loads (a test of the flags first would put a memory round trip between them)
    const int f1 = d.cg_flag[1], f0 = d.cg_flag[0];
    const double tol2 = CG_TOL2(d), bb = d.cg_scal[2];
    double rr_prev = 0, pq = 0, rz = 0;
    if (d.cg_two_level) { rr_prev = *CG_FIN_RR(d, par ^ 1); rz = *CG_FIN_RZ(d, par ^ 1); pq = *CG_FIN_PQ(d); }
    if (f1 || f0) return;
    if (!d.cg_two_level) {
        for (int t = threadIdx.x; t < d.cg_nparts; t += 256) { rr_prev += CG_RR(d, par ^ 1)[t]; rz += CG_RZ(d, par ^ 1)[t]; }
        for (int t = threadIdx.x; t < d.cg_nparts_spmv; t += 256) pq += CG_PQ(d)[t];
        block_sum3_256(rr_prev, pq, rz, red + 8);
    }
    if (rr_prev <= tol2 * bb) return;
    if (!(pq > 0)) { if (blockIdx.x == 0 && threadIdx.x == 0) d.cg_flag[1] = 1; return; }
    const double alpha = rz / pq;
    double* rnew = d.cg_r[par ^ 1];
    double* my_rn = pc_rn + (size_t)wave * N; double* y = yw + (size_t)wave * N;
#pragma unroll
    for (int u = 0; u < NV; u++) {                                            // the whole block's new residual, once per wavefront
        const int t = lane + 64 * u, i = row0 + t;
        if (t < N) {
            const double v = i < d.sp ? v_r[u] - alpha * v_q[u] : 0.0;
            my_rn[t] = v; y[t] = 0.0;
            if (wave == 0 && i < d.sp) { d.x[i] += alpha * v_p[u]; rnew[i] = v; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane l holds row l % 16, columns 4 (l / 16) .. + 3 of its tile (ba_pc_pack_kernel): the column sums over the 16 rows then run inside a DPP row (xor 1, 2, 4, 8:
    // vector-ALU moves), the row sums over the four column groups across the rows (xor 16, 32: the two exchanges left on the LDS crossbar).  Round 5's map (row l / 4,
    // columns 4 (l % 4)) put the sixteen-row sums of FOUR values on xor 4 .. 32; the pairing order of both sums is unchanged (row bit 0 first; group bit 0 first): same bits.
    const int row = lane & 15, c4 = lane >> 4;
#pragma unroll
    for (int u = 0; u < TPW; u++) {
        const int ti = wave + 4 * u;
        if (ti < T) {
            int I = 0, base = 0;
            while (base + (NT - I) <= ti) { base += NT - I; I++; }
            const int J = I + (ti - base);
            const double* rJ = my_rn + 16 * J + 4 * c4;
            const double rI = my_rn[16 * I + row];
            const double m0 = (double)m[u].x, m1 = (double)m[u].y, m2 = (double)m[u].z, m3 = (double)m[u].w;
            double rp = m0 * rJ[0] + m1 * rJ[1] + m2 * rJ[2] + m3 * rJ[3];
            rp = lx_xadd16(rp, rp); rp = lx_xadd32(rp, rp);
            if (c4 == 0) y[16 * I + row] += rp;
            if (I != J) {
                double c0 = m0 * rI, c1 = m1 * rI, c2 = m2 * rI, c3 = m3 * rI;
                c0 += lx_xor<1>(c0); c1 += lx_xor<1>(c1); c2 += lx_xor<1>(c2); c3 += lx_xor<1>(c3);
                c0 += lx_xor<2>(c0); c1 += lx_xor<2>(c1); c2 += lx_xor<2>(c2); c3 += lx_xor<2>(c3);
                c0 += lx_xor<4>(c0); c1 += lx_xor<4>(c1); c2 += lx_xor<4>(c2); c3 += lx_xor<4>(c3);
                c0 += lx_xor<8>(c0); c1 += lx_xor<8>(c1); c2 += lx_xor<8>(c2); c3 += lx_xor<8>(c3);
                if (row == 0) { double* yj = y + 16 * J + 4 * c4; yj[0] += c0; yj[1] += c1; yj[2] += c2; yj[3] += c3; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    int tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__shfl(tk, 0) != 3) return;
    double rzn = 0, rrn = 0;
    for (int t = lane; t < N; t += 64) {
        const int i = row0 + t;
        const double z = ((yw[t] + yw[N + t]) + yw[2 * N + t]) + yw[3 * N + t];
        if (i < d.sp) { d.cg_z[i] = z; rzn += my_rn[t] * z; rrn += my_rn[t] * my_rn[t]; }
    }
    rzn = lx_wave_sum(rzn); rrn = lx_wave_sum(rrn);
    if (lane == 0) { cg_publish(&CG_RZ(d, par)[blockIdx.x], rzn); cg_publish(&CG_RR(d, par)[blockIdx.x], rrn); }
    if (d.cg_two_level)
        cg_tree_reduce(d.cg_tick, CG_TICK3(d, 0), d.cg_ngrp, CG_RZ(d, par), CG_RR(d, par), CG2_RZ(d, par), nullptr, CG2_RR(d, par), nullptr,
                       CG_FIN_RZ(d, par), nullptr, CG_FIN_RR(d, par), nullptr, d.cg_nparts);
}
// pc_pack32 from pc_inv32: one workgroup per block, a thread per element of a tile
__global__ __launch_bounds__(256) void ba_pc_pack_kernel(CorbBADev d)
{
    const int b = blockIdx.x, n = d.pc_gb, nt = n / 16, t = threadIdx.x, row = t >> 4, c = t & 15;
    const float* sq = d.pc_inv32 + (size_t)b * n * n;
    float* out = d.pc_pack32 + (size_t)b * (nt * (nt + 1) / 2) * 256;
    int ti = 0;
    // element (row, c) of a tile goes where lane (c / 4) * 16 + row finds it as component c % 4 of its float4 (see ba_pcg_step_sym_body)
    const int pos = 4 * ((c >> 2) * 16 + row) + (c & 3);
    for (int I = 0; I < nt; I++) for (int J = I; J < nt; J++, ti++) out[ti * 256 + pos] = sq[(size_t)(16 * I + row) * n + 16 * J + c];
}
__global__ __launch_bounds__(256) void ba_pcg_step_big_kernel(CorbBADev d, int par)
{
    __shared__ double red[20];
    __shared__ int cnt;
    __shared__ double pc_yw[4 * 96];
    extern __shared__ double pc_rn[];                      // [4][pc_gb]
    if (d.pc_pack32) { if (d.pc_gb == 96) ba_pcg_step_sym_body<6>(d, par, pc_rn, pc_yw, red, &cnt); else ba_pcg_step_sym_body<3>(d, par, pc_rn, pc_yw, red, &cnt); }
    else
    if (d.pc_inv32) ba_pcg_step_big_body<float>(d, d.pc_inv32, par, pc_rn, red, &cnt);
    else ba_pcg_step_big_body<double>(d, d.pc_inv, par, pc_rn, red, &cnt);
}

// bb = |b|^2, iteration counter, x = 0 (b_schur has been consumed)
__global__ __launch_bounds__(256) void ba_pcg_zero_x_kernel(CorbBADev d, double tol2)
{
    __shared__ double red[4];
    // workgroup 0 files the scalars; every workgroup clears its stride of x (one workgroup took 41 us for 300 000 entries at 50 000 keyframes)
    if (blockIdx.x == 0) {
        const double bb = cg_reduce_parts(CG_RR(d, 1), d.cg_nparts, red);
        if (threadIdx.x == 0) { d.cg_scal[2] = bb; d.cg_scal[3] = bb; d.cg_scal[4] = 0; CG_TOL2(d) = tol2; if (!(bb > 0)) d.cg_flag[0] = 1; }
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.sp; i += gridDim.x * 256) d.x[i] = 0.0;
}

// bsr_tslot[s] = s for a block on / above the diagonal, else the slot of its transpose (k, j) -> (j, k) in row j: where the SpMV reads a lower block from
__global__ __launch_bounds__(256) void ba_tslot_kernel(CorbBADev d, int* tslot)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d.nP) return;
    for (int s = d.bsr_rowptr[k]; s < d.bsr_rowptr[k + 1]; s++) {
        const int j = d.bsr_col[s];
        int a = s;
        if (j < k) {
            a = d.bsr_rowptr[j]; int b = d.bsr_rowptr[j + 1] - 1;
            while (a < b) { const int mid = (a + b) >> 1; if (d.bsr_col[mid] < k) a = mid + 1; else b = mid; }
        }
        tslot[s] = a;
    }
}
void ba_launch_tslot(const CorbBADev& d, int* tslot, hipStream_t s) { if (d.nP > 0) hipLaunchKernelGGL(ba_tslot_kernel, dim3((d.nP + 255) / 256), dim3(256), 0, s, d, tslot); }
// iteration t (parity par = t & 1): beta = rz_t / rz_{t-1}; p_t = z + beta p_{t-1}; q = S p_t; partial p.q
// (t = 0: p_{-1} = 0 and both rz slots hold rz_0, so beta = 1 multiplies zeros)
__global__ __launch_bounds__(256) void ba_pcg_spmv_kernel(CorbBADev d, int par)
{
    // one wavefront per block row: 10 lane groups x 6 rows sweep the row's 6x6 blocks 10 at a time.  No workgroup barrier after the first instructions
    // (cg_wave_handoff), and the first trip's operands that do not depend on the iteration's scalars are requested before those are read.
    __shared__ double red[20];
    __shared__ double spmv_q[256];
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();                                      // (before the first load: a barrier also waits for the wavefront's outstanding loads)
    // Round 4: the matrix is symmetric and a block (k, j < k) was streamed a moment ago as block (j, k) of row j -- by the SAME XCD when rows are dealt out in
    // contiguous eighths (workgroup b runs on XCD b % 8 and takes rows of the (b % 8)-th eighth): the lower blocks are read through bsr_tslot from the upper
    // triangle, transposed (six 8-byte loads per lane instead of three 16-byte ones), and come out of that XCD's L2; HBM serves half the bytes.
    const int per = (int)gridDim.x >> 3;                    // (the grid is a multiple of 8 workgroups)
    // (Round 5, measured and dropped: the rows handed out in an order that interleaves the eight clients' trajectories by position -- an XCD's rows a PLACE instead of a
    // client, so that the inter-client half of the lower blocks would meet its upper twin in the same L2: 86.6 -> 105.8 ms of solve per 10 LM iterations at 50 000 keyframes.)
    const int lane = threadIdx.x & 63, k = (((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3)) * 4 + (threadIdx.x >> 6);
    const int grp = lane / 6, a = lane - 6 * grp;
    const bool act = k < d.nP && lane < 60;
    int s = 0, s_end = 0, j = 0, jn = 0, tn = 0;
    double S0[6] = { 0, 0, 0, 0, 0, 0 }, Z0[6] = { 0, 0, 0, 0, 0, 0 };
    // row a of block s: straight from the upper triangle, or column a of its transpose
#define SPMV_LOAD(dst, slot, tslot) do { if ((tslot) == (slot)) { const double* Sv_ = d.bsr_val + (size_t)(slot) * 36 + a * 6; _Pragma("unroll") for (int c = 0; c < 6; c++) dst[c] = Sv_[c]; } \
        else { const double* Tv_ = d.bsr_val + (size_t)(tslot) * 36 + a; _Pragma("unroll") for (int c = 0; c < 6; c++) dst[c] = Tv_[6 * c]; } } while (0)
    if (act) {
        s = d.bsr_rowptr[k] + grp; s_end = d.bsr_rowptr[k + 1];
        if (s < s_end) {
            const int sn = s + 10 < s_end ? s + 10 : s;
            j = d.bsr_col[s]; jn = d.bsr_col[sn]; tn = d.bsr_tslot[sn];
            const int t0 = d.bsr_tslot[s];
            SPMV_LOAD(S0, s, t0);
            const double* zj0 = d.cg_z + 6 * (size_t)j;          // (z does not depend on the iteration's scalars any more: the first trip's gather goes out with its block)
#pragma unroll
            for (int c = 0; c < 6; c++) Z0[c] = zj0[c];
        }
    }
    // the flags and the iteration's scalars as one batch of loads behind the operands (a test of the flags first puts a memory round trip between them)
    const int f1 = d.cg_flag[1], f0 = d.cg_flag[0];
    const double tol2 = CG_TOL2(d), bb = d.cg_scal[2];
    double rr = 0, rz_new = 0, rz_old = 0;
    if (d.cg_two_level) { rr = *CG_FIN_RR(d, par ^ 1); rz_new = *CG_FIN_RZ(d, par ^ 1); rz_old = *CG_FIN_RZ(d, par); }
    if (f1 || f0) return;                                 // failed, or converged in an EARLIER kernel (the r.r slot of the other parity is stale then)
    if (!d.cg_two_level) {
        for (int t = threadIdx.x; t < d.cg_nparts; t += 256) { rr += CG_RR(d, par ^ 1)[t]; rz_new += CG_RZ(d, par ^ 1)[t]; rz_old += CG_RZ(d, par)[t]; }
        block_sum3_256(rr, rz_new, rz_old, red + 8);
    }
    if (rr <= tol2 * bb) { if (blockIdx.x == 0 && threadIdx.x == 0) { d.cg_flag[0] = 1; d.cg_scal[3] = rr; } return; }   // converged
    const double beta = rz_new / rz_old;
    const double* pold = d.cg_p[par ^ 1]; double* pnew = d.cg_p[par];
    double q = 0;
    if (act) {
        // Round 3, measured at 50 000 keyframes (105 us per launch, 430 MB) and dropped: (1) column indices by one coalesced load + shuffles and three
        // blocks in flight per lane group (27 independent 16-byte loads, 108 VGPRs): 22 % slower -- the kernel lives on wavefronts in flight, not on
        // loads per wavefront; (2) XCD-aware rows (XCD x takes the x-th eighth of the block rows, so that its L2 holds one eighth of z / p): no change;
        // (3) 2 / 4 / 8 block rows per wavefront (the partial-sum prologue and the group ticket paid once per 8 / 16 / 32 rows): 0 / +5 / +12 %;
        // (3b) two blocks per lane group in flight (both index loads, then both operand sets): +5 % -- every form with MORE requests per wavefront lost;
        // (4) a single-precision copy of the blocks in the recurrence (202 instead of 403 MB: 105 -> 71 us per launch, solve 333 -> 263 ms) -- but the true
        // residual of its solution stalls at 3e-7 .. 3e-6 |b|, and the refinement rounds that bring it to the 1e-8 of the all-double solve (restart from the
        // double-precision residual) need 40 % more iterations: 349 ms.  Accepting 3e-6 would have been 2.3e-6 in chi2 -- inside the parity bar, but a
        // tolerance that depends on the map size is not what g2o's exact solve does.
        // Round 5: q_t = S z_t + beta q_{t-1} (S p_t = S (z_t + beta p_{t-1}), and S p_{t-1} is the q this kernel filed an iteration ago -- the Chronopoulos / Gear form of
        // the recurrence): a block's lanes gather ONE vector (z_j) instead of two (z_j and p_{t-1,j}) -- the kernel is bound by its request stream, and the vector gathers
        // were a third of it.  p_t itself is still formed (own rows only) for the update of x.
        if (s < s_end) {
#pragma unroll
            for (int c = 0; c < 6; c++) q += S0[c] * Z0[c];
        }
        // (the NEXT trip's column index is requested a trip ahead: a trip is then one memory round trip -- operands -- instead of two -- index, operands)
        for (s += 10; s < s_end; s += 10) {
            const int jj = jn, tt = tn;
            const int sn = s + 10 < s_end ? s + 10 : s;
            jn = d.bsr_col[sn]; tn = d.bsr_tslot[sn];       // (one packed {column, tslot} word per slot instead of two loads: 80.9 -> 82.2 ms of solve, round 5: dropped)
            double Sv[6];
            SPMV_LOAD(Sv, s, tt);
            const double* zj = d.cg_z + 6 * (size_t)jj;
#pragma unroll
            for (int c = 0; c < 6; c++) q += Sv[c] * zj[c];
        }
#undef SPMV_LOAD
    }
    // the ten lane groups' sums of row a, added in group order (fixed order => deterministic): through the wavefront's own 512 bytes of LDS -- one 8-byte store per
    // lane, nine 8-byte loads on lanes 0 .. 5 -- instead of twenty ds_bpermute_b32 on every lane
    double qt = q;
    {
        double* qs = spmv_q + 64 * (threadIdx.x >> 6);
        qs[lane] = q;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 6) {
#pragma unroll
            for (int m = 1; m < 10; m++) qt += qs[lane + 6 * m];
        }
    }
    double pq = 0;
    if (k < d.nP && lane < 6) {
        const size_t i = 6 * (size_t)k + lane;
        const double pi = d.cg_z[i] + beta * pold[i];
        const double qi = qt + beta * d.cg_q[i];
        pnew[i] = pi; d.cg_q[i] = qi;
        pq = pi * qi;
    }
    // the wavefront's p.q: only lanes 0 .. 5 hold a term, so the stages 32, 16, 8 of the wave sum add zeros -- the last three (DPP moves) give lane 0 the same bits
    pq += lx_xor<4>(pq); pq += lx_xor<2>(pq); pq += lx_xor<1>(pq);
    {
        int tk = 0;
        if (lane == 0) { red[threadIdx.x >> 6] = pq; tk = __hip_atomic_fetch_add(&cnt, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (__builtin_amdgcn_readfirstlane(tk) != 3) return;      // (as cg_wave_handoff: the last of the four wavefronts adds their sums in wavefront order)
        pq = red[0] + red[1] + red[2] + red[3];
    }
    if (lane == 0) cg_publish(&CG_PQ(d)[blockIdx.x], pq);
    if (blockIdx.x == 0 && lane == 0) d.cg_scal[4] += 1.0;
    if (d.cg_two_level)
        cg_tree_reduce(d.cg_tick + (size_t)d.cg_ngrp * CG_TICK_STRIDE, CG_TICK3(d, 1), d.cg_ngrp_spmv, CG_PQ(d), nullptr, CG2_PQ(d), nullptr, nullptr, nullptr,
                       CG_FIN_PQ(d), nullptr, nullptr, nullptr);
}

// alpha = rz_t / (p.q); x += alpha p; r_{t+1} = r_t - alpha q; z = Minv r_{t+1}; partial r.z, r.r into slot par
__global__ __launch_bounds__(256) void ba_pcg_step_kernel(CorbBADev d, int par)
{
    __shared__ double red[12];
    if (d.cg_flag[1] || d.cg_flag[0]) return;             // failed, or converged in an EARLIER kernel (the r.r slot of the other parity is stale then)
    double rr_prev = 0, pq = 0, rz = 0;
    if (d.cg_two_level) { rr_prev = *CG_FIN_RR(d, par ^ 1); rz = *CG_FIN_RZ(d, par ^ 1); pq = *CG_FIN_PQ(d); }
    else {
        for (int t = threadIdx.x; t < d.cg_nparts; t += 256) { rr_prev += CG_RR(d, par ^ 1)[t]; rz += CG_RZ(d, par ^ 1)[t]; }
        for (int t = threadIdx.x; t < d.cg_nparts_spmv; t += 256) pq += CG_PQ(d)[t];
        block_sum3_256(rr_prev, pq, rz, red);
    }
    if (rr_prev <= CG_TOL2(d) * d.cg_scal[2]) return;                               // converged: spmv of this iteration did not run
    if (!(pq > 0)) { if (blockIdx.x == 0 && threadIdx.x == 0) d.cg_flag[1] = 1; return; }    // not positive definite
    const double alpha = rz / pq;
    const double* p = d.cg_p[par];
    const double* rold = d.cg_r[par]; double* rnew = d.cg_r[par ^ 1];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double rzn = 0, rrn = 0;
    if (i < d.sp) {
        const int k = i / 6, a = i % 6;
        double rn[6];
#pragma unroll
        for (int c = 0; c < 6; c++) rn[c] = rold[6 * (size_t)k + c] - alpha * d.cg_q[6 * (size_t)k + c];
        const double* Mi = d.Minv + (size_t)k * 36 + a * 6;
        const double z = Mi[0] * rn[0] + Mi[1] * rn[1] + Mi[2] * rn[2] + Mi[3] * rn[3] + Mi[4] * rn[4] + Mi[5] * rn[5];
        d.x[i] += alpha * p[i];
        rnew[i] = rn[a]; d.cg_z[i] = z;                  // z is not read in this kernel; r is double-buffered
        rzn = rn[a] * z; rrn = rn[a] * rn[a];
    }
    double dummy = 0;
    block_sum3_256(rzn, rrn, dummy, red);
    if (threadIdx.x == 0) { cg_publish(&CG_RZ(d, par)[blockIdx.x], rzn); cg_publish(&CG_RR(d, par)[blockIdx.x], rrn); }
    if (d.cg_two_level && threadIdx.x < 64)
        cg_tree_reduce(d.cg_tick, CG_TICK3(d, 0), d.cg_ngrp, CG_RZ(d, par), CG_RR(d, par), CG2_RZ(d, par), nullptr, CG2_RR(d, par), nullptr,
                       CG_FIN_RZ(d, par), nullptr, CG_FIN_RR(d, par), nullptr);
}

// after the last enqueued iteration: publish convergence (the test otherwise happens at the next spmv)
__global__ __launch_bounds__(256) void ba_pcg_check_kernel(CorbBADev d, int par_last)
{
    __shared__ double red[4];
    const double rr = cg_reduce_parts(CG_RR(d, par_last), d.cg_nparts, red);
    if (threadIdx.x == 0) { d.cg_scal[3] = rr; if (rr <= CG_TOL2(d) * d.cg_scal[2]) d.cg_flag[0] = 1; }
}

// ------------------------------------------------------------------------------------------------
// Self-certification of the reduced solve (round 5).  g2o's LinearSolverEigen is an exact factorisation (G/solvers/linear_solver_eigen.h:94-124): its
// x satisfies S x = b to rounding.  The PCG solve stops on the RECURRENCE residual, which drifts from the true one over hundreds of iterations; so after
// every solve the true residual |b - S x| / |b| is recomputed in FP64 by a kernel that shares nothing with the CG kernels but the matrix -- a thread per
// scalar row walking its block row, the lower blocks as the transposes of their stored twins -- against a copy of b taken before the solve consumed it.
// CorbBAResult.pcg_residual_max / _last carry it out; tests/test_gpu_ba.py asserts it at 50 000 keyframes.  (~0.1 ms per solve at that size.)
__global__ __launch_bounds__(256) void ba_true_residual_kernel(CorbBADev d, const double* __restrict__ b, double* part)
{
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double rr = 0, bb = 0;
    if (i < d.sp) {
        const int k = i / 6, a = i - 6 * k;
        double acc = b[i];
        for (int s = d.bsr_rowptr[k]; s < d.bsr_rowptr[k + 1]; s++) {
            const int t = d.bsr_tslot[s];
            const double* xj = d.x + 6 * (size_t)d.bsr_col[s];
            const double* B = d.bsr_val + (size_t)t * 36;
            if (t == s) { for (int c = 0; c < 6; c++) acc -= B[a * 6 + c] * xj[c]; }
            else { for (int c = 0; c < 6; c++) acc -= B[c * 6 + a] * xj[c]; }
        }
        rr = acc * acc; bb = b[i] * b[i];
    }
    const double s1 = block_sum_256(rr, red);
    const double s2 = block_sum_256(bb, red);
    if (threadIdx.x == 0) { part[blockIdx.x] = s1; part[gridDim.x + blockIdx.x] = s2; }
}
// out[0] = the largest relative residual of the call's solves so far (a NaN stays), out[1] = this solve's
__global__ __launch_bounds__(256) void ba_true_residual_fin_kernel(const double* part, int n, double* out)
{
    __shared__ double red[4];
    const double rr = cg_reduce_parts(part, n, red);
    const double bb = cg_reduce_parts(part + n, n, red);
    if (threadIdx.x == 0) { const double v = bb > 0 ? sqrt(rr / bb) : 0.0; out[1] = v; if (!(v <= out[0])) out[0] = v; }
}
void ba_launch_true_residual(const CorbBADev& d, const double* b, double* part, double* out, hipStream_t s)
{
    const int n = (d.sp + 255) / 256;
    hipLaunchKernelGGL(ba_true_residual_kernel, dim3(n), dim3(256), 0, s, d, b, part);
    hipLaunchKernelGGL(ba_true_residual_fin_kernel, dim3(1), dim3(256), 0, s, part, n, out);
}
// |v|_inf of n doubles into *out (zeroed by the caller): non-negative doubles order like their bit patterns, so the maximum is an integer atomicMax --
// order-independent, hence deterministic.  A NaN (exponent all ones) wins, as it should.  Used for |J'r|_inf = |b|_inf at the estimates a call returns.
__global__ __launch_bounds__(256) void ba_absmax_kernel(const double* __restrict__ v, size_t n, unsigned long long* out)
{
    unsigned long long m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(fabs(v[i]));
        m = u > m ? u : m;
    }
    m = lx_wave_max_u64(m);
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
void ba_launch_absmax(const double* v, size_t n, double* out, hipStream_t s)
{
    (void)hipMemsetAsync(out, 0, sizeof(double), s);
    if (n == 0) return;
    const int nb = (int)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(ba_absmax_kernel, dim3(nb), dim3(256), 0, s, v, n, reinterpret_cast<unsigned long long*>(out));
}

// ------------------------------------------------------------------------------------------------
// Deterministic Schur complement on the FP64 matrix cores (block_solver.hpp:400-431), no atomics.
// Structure (once per optimize() call): for every block (p, q >= p) of the pattern the two poses' landmark lists (ascending) are merged into the
// list of edge pairs (e1 = edge (p,l), e2 = edge (q,l)); per LM trial V_e = W_e C_l (C_l C_l' = (Hll_l + lambda I)^-1) is formed once per edge and
//     S(p,q) = [p==q](Hpp + lambda I) - sum_pairs V_e1 V_e2'
// is ONE contraction of depth 3 x pairs per block: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction), the sum stays in
// the accumulator registers, the block and its transpose are stored once -- bit-identical from run to run.  Lane maps of the instruction
// (probed on gfx950, tools/ubench/mfma_f64.hip / profiles/r02_ubench): A[blk][i][k] at lane 16k + 4blk + i, B[blk][k][j] at lane 16k + 4blk + j,
// D[blk][i][j] at lane 16i + 4blk + j.  FP64 MFMA and FP64 FMA have the same peak on this part (78 TFLOP/s, same file): the matrix instruction is
// used because it needs a quarter of the operand loads per multiply-add, and operand gathering is what bounds this kernel.
// History at 10 000 keyframes / 3.19 M observations (profiles/r02_ba_*): LDS-atomic row-owner kernel 1.86 ms -> quadrant blocks, one chunk at a
// time 1.05 -> grouped loads 0.89 -> XCD-aware block order 0.77 -> contraction split over the four blocks 0.70 -> one operand array (V) 0.57 ms.
__device__ __forceinline__ int ba_merge_pairs(const CorbBADev& d, int p, int q, int2* out)
{
    int i = d.poff[p], j = d.poff[q], n = 0;
    const int ie = d.poff[p + 1], je = d.poff[q + 1];
    int la = i < ie ? d.plm[i] : -1, lb = j < je ? d.plm[j] : -1;      // free landmarks ascending, then the -1s of the fixed ones
    while (la >= 0 && lb >= 0) {
        if (la == lb) {
            // edge i pairs with EVERY edge of q on this landmark, and j stays at the start of that run for the next edge of p: a (keyframe, map point)
            // observation that occurs twice contributes (W_e + W_e') Dinv (...)' -- all the cross products -- exactly like the summed Hpl block of g2o
            for (int j2 = j; j2 < je && d.plm[j2] == la; j2++) { if (out) out[n] = make_int2(d.row_schur ? i - d.poff[p] : d.pedge[i], d.v_kf ? j2 : d.pedge[j2]); n++; }
            i++; la = i < ie ? d.plm[i] : -1;
        } else if (la < lb) { i++; la = i < ie ? d.plm[i] : -1; }
        else { j++; lb = j < je ? d.plm[j] : -1; }
    }
    return n;
}
__global__ __launch_bounds__(256) void ba_pairs_count_kernel(CorbBADev d)
{
    const int u = blockIdx.x * 256 + threadIdx.x;            // upper block (p, q >= p) number u; uinfo = (slot, p, q, slot of the transposed block)
    if (u >= d.nu) return;
    const int4 in = d.uinfo[u];
    const int p = in.y, q = in.z;
    int m = in.x;
    if (q != p) {                                             // slot of (q, p)
        int a = d.bsr_rowptr[q], b = d.bsr_rowptr[q + 1] - 1;
        while (a < b) { const int mid = (a + b) >> 1; if (d.bsr_col[mid] < p) a = mid + 1; else b = mid; }
        m = a;
    }
    d.pair_off[u] = ba_merge_pairs(d, p, q, nullptr); d.uinfo[u].w = m;
}
// exclusive scan of pair_off[0 .. nu) in place, total into pair_off[nu]: one workgroup, a contiguous chunk per thread
__global__ __launch_bounds__(1024) void ba_pairs_scan_kernel(CorbBADev d)
{
    __shared__ int part[1024];
    const int n = d.nu, t = threadIdx.x, chunk = (n + 1023) / 1024;
    const int i0 = min(n, t * chunk), i1 = min(n, i0 + chunk);
    int sum = 0;
    for (int i = i0; i < i1; i++) sum += d.pair_off[i];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { const int v = t >= o ? part[t - o] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    int run = part[t] - sum;
    for (int i = i0; i < i1; i++) { const int c = d.pair_off[i]; d.pair_off[i] = run; run += c; }
    if (t == 1023) d.pair_off[n] = part[1023];
}
__global__ __launch_bounds__(256) void ba_pairs_fill_kernel(CorbBADev d)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= d.nu) return;
    const int4 in = d.uinfo[u];
    if (d.pair_off[u + 1] > d.pair_off[u]) (void)ba_merge_pairs(d, in.y, in.z, const_cast<int2*>(d.pairs) + d.pair_off[u]);
}
// Few blocks with long lists (a local window: 15 blocks of 2 000 landmarks each): one WAVEFRONT per block instead of one thread.  Lane l owns the
// l-th 64th of p's list and merges it serially against q's list from the lower bound of its first landmark on; the per-lane counts are
// prefix-summed, so the pairs come out in the same ascending-landmark order as ba_merge_pairs'.
__device__ __forceinline__ int ba_plm_valid(const CorbBADev& d, int b, int e)           // first index in [b, e) with plm < 0 (fixed landmarks sit at the end)
{ while (b < e) { const int mid = (b + e) >> 1; if (d.plm[mid] >= 0) b = mid + 1; else e = mid; } return b; }
__device__ __forceinline__ int ba_merge_chunk(const CorbBADev& d, int ia, int i0, int i1, int jb, int je, int2* out)      // ia = poff[p]: the start of p's list
{
    if (i0 >= i1 || jb >= je) return 0;
    const int first = d.plm[i0];
    int lo = jb, hi = je;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (d.plm[mid] < first) lo = mid + 1; else hi = mid; }
    int i = i0, j = lo, n = 0;
    while (i < i1 && j < je) {
        const int la = d.plm[i], lb = d.plm[j];
        if (la == lb) { for (int j2 = j; j2 < je && d.plm[j2] == la; j2++) { if (out) out[n] = make_int2(d.row_schur ? i - ia : d.pedge[i], d.v_kf ? j2 : d.pedge[j2]); n++; } i++; }      // (see ba_merge_pairs)
        else if (la < lb) i++;
        else j++;
    }
    return n;
}
__global__ __launch_bounds__(256) void ba_pairs_wave_kernel(CorbBADev d, int fill)
{
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= d.nu) return;
    const int4 in = d.uinfo[u];
    const int p = in.y, q = in.z;
    const int ia = d.poff[p], ie = ba_plm_valid(d, ia, d.poff[p + 1]), jb = d.poff[q], je = ba_plm_valid(d, jb, d.poff[q + 1]);
    const int chunk = (ie - ia + 63) >> 6;
    const int i0 = min(ia + lane * chunk, ie), i1 = min(i0 + chunk, ie);
    if (fill && d.pair_off[u + 1] == d.pair_off[u]) return;
    const int c = ba_merge_chunk(d, ia, i0, i1, jb, je, nullptr);
    int incl = c;
    incl = lx_wave_incl_scan_i(incl);
    if (!fill) {
        if (lane == 63) d.pair_off[u] = incl;
        if (lane == 0) {
            int m = in.x;
            if (q != p) {                                         // slot of (q, p)
                int a = d.bsr_rowptr[q], b = d.bsr_rowptr[q + 1] - 1;
                while (a < b) { const int mid = (a + b) >> 1; if (d.bsr_col[mid] < p) a = mid + 1; else b = mid; }
                m = a;
            }
            d.uinfo[u].w = m;
        }
        return;
    }
    if (c > 0) (void)ba_merge_chunk(d, ia, i0, i1, jb, je, const_cast<int2*>(d.pairs) + d.pair_off[u] + (incl - c));
}
// A local window has a handful of blocks (15 for 5 free keyframes) with ~2 000 landmarks each: a WORKGROUP per block, thread t owning the t-th 256th of p's
// list (a wavefront per block left the window's four launches at 68 us each); same pairs in the same order.
__global__ __launch_bounds__(256) void ba_pairs_block_kernel(CorbBADev d, int fill)
{
    __shared__ int wtot[4];
    const int u = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int4 in = d.uinfo[u];
    const int p = in.y, q = in.z;
    const int ia = d.poff[p], ie = ba_plm_valid(d, ia, d.poff[p + 1]), jb = d.poff[q], je = ba_plm_valid(d, jb, d.poff[q + 1]);
    const int chunk = (ie - ia + 255) >> 8;
    const int i0 = min(ia + t * chunk, ie), i1 = min(i0 + chunk, ie);
    if (fill && d.pair_off[u + 1] == d.pair_off[u]) return;
    const int c = ba_merge_chunk(d, ia, i0, i1, jb, je, nullptr);
    int incl = c;
    incl = lx_wave_incl_scan_i(incl);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; w++) before += wtot[w];
    incl += before;
    if (!fill) {
        if (t == 255) d.pair_off[u] = incl;
        if (t == 0) {
            int m = in.x;
            if (q != p) {                                         // slot of (q, p)
                int a = d.bsr_rowptr[q], b = d.bsr_rowptr[q + 1] - 1;
                while (a < b) { const int mid = (a + b) >> 1; if (d.bsr_col[mid] < p) a = mid + 1; else b = mid; }
                m = a;
            }
            d.uinfo[u].w = m;
        }
        return;
    }
    if (c > 0) (void)ba_merge_chunk(d, ia, i0, i1, jb, je, const_cast<int2*>(d.pairs) + d.pair_off[u] + (incl - c));
}
// Maps (the row-owner Schur path: urow is filed): a WORKGROUP per block row p.  The row's landmark list goes into an LDS hash table once (landmark -> first position in
// the list | run length << 16), then a wavefront per block (p, q) streams q's list -- 64 consecutive entries per load -- and PROBES: the entry that opens a run of
// q's list on a landmark of p owns that landmark's (run of p) x (run of q) pairs, a wavefront prefix sum over the pair counts gives every lane its output offset,
// i.e. the pairs come out landmark by landmark, p's position major, exactly in the order of the two-list merge.  A block costs its q list's loads instead of the
// ~1 100 dependent steps of a serial merge per thread (3.8 ms to count + 8.3 ms to fill the 97 M pairs of a 50 000-keyframe map).  Rows with more than
// BA_PAIRS_HASH_MAX observations of free landmarks (hub keyframes) are merged serially, a thread per block.
#define BA_PAIRS_HASH_SLOTS 4096
#define BA_PAIRS_HASH_MAX 2048
__global__ __launch_bounds__(256) void ba_pairs_row_kernel(CorbBADev d, int fill)
{
    __shared__ int hkey[BA_PAIRS_HASH_SLOTS];
    __shared__ int hval[BA_PAIRS_HASH_SLOTS];
    const int p = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int u0 = d.urow[p], u1 = d.urow[p + 1];
    if (u1 <= u0) return;
    const int ia = d.poff[p], ie = ba_plm_valid(d, ia, d.poff[p + 1]);
    auto transposed_slot = [&](int u, int q, int m) {
        if (q != p) {                                         // slot of (q, p)
            int a = d.bsr_rowptr[q], b = d.bsr_rowptr[q + 1] - 1;
            while (a < b) { const int mid = (a + b) >> 1; if (d.bsr_col[mid] < p) a = mid + 1; else b = mid; }
            m = a;
        }
        d.uinfo[u].w = m;
    };
    if (ie - ia > BA_PAIRS_HASH_MAX) {                        // a hub keyframe: the serial merge, a thread per block
        for (int u = u0 + t; u < u1; u += 256) {
            const int4 in = d.uinfo[u];
            const int q = in.z, jb = d.poff[q], je = ba_plm_valid(d, jb, d.poff[q + 1]);
            if (!fill) { d.pair_off[u] = ba_merge_chunk(d, ia, ia, ie, jb, je, nullptr); transposed_slot(u, q, in.x); }
            else if (d.pair_off[u + 1] > d.pair_off[u]) (void)ba_merge_chunk(d, ia, ia, ie, jb, je, const_cast<int2*>(d.pairs) + d.pair_off[u]);
        }
        return;
    }
    for (int k = t; k < BA_PAIRS_HASH_SLOTS; k += 256) hkey[k] = -1;
    __syncthreads();
    for (int i = ia + t; i < ie; i += 256) {
        const int l = d.plm[i];
        if (i > ia && d.plm[i - 1] == l) continue;            // a run's first entry files it
        int run = 1; while (i + run < ie && d.plm[i + run] == l) run++;
        unsigned h = ((unsigned)l * 2654435761u) >> 20;        // 12 bits
        for (;;) { const int prev = atomicCAS(&hkey[h], -1, l); if (prev == -1) { hval[h] = (i - ia) | (run << 16); break; } h = (h + 1) & (BA_PAIRS_HASH_SLOTS - 1); }
    }
    __syncthreads();
    for (int u = u0 + wave; u < u1; u += 4) {
        const int4 in = d.uinfo[u];
        const int q = in.z, jb = d.poff[q], je = ba_plm_valid(d, jb, d.poff[q + 1]);
        if (fill && d.pair_off[u + 1] == d.pair_off[u]) continue;
        int2* out = fill ? const_cast<int2*>(d.pairs) + d.pair_off[u] : nullptr;
        int total = 0;
        for (int c0 = jb; c0 < je; c0 += 64) {
            const int j = c0 + lane;
            int cnt = 0, pos = 0, runp = 0, runq = 0;
            if (j < je) {
                const int l = d.plm[j];
                if (j == jb || d.plm[j - 1] != l) {           // opens a run of q's list
                    unsigned h = ((unsigned)l * 2654435761u) >> 20;
                    for (;;) { const int k = hkey[h]; if (k == l) { const int v = hval[h]; pos = v & 0xFFFF; runp = v >> 16; break; } if (k == -1) break; h = (h + 1) & (BA_PAIRS_HASH_SLOTS - 1); }
                    if (runp) { runq = 1; while (j + runq < je && d.plm[j + runq] == l) runq++; cnt = runp * runq; }
                }
            }
            int incl = cnt;
            incl = lx_wave_incl_scan_i(incl);
            if (fill && cnt) {
                int2* o = out + total + (incl - cnt);
                for (int a = 0; a < runp; a++) for (int b = 0; b < runq; b++) *o++ = make_int2(d.row_schur ? pos + a : d.pedge[ia + pos + a], d.v_kf ? j + b : d.pedge[j + b]);
            }
            total += __shfl(incl, 63);
        }
        if (!fill && lane == 0) { d.pair_off[u] = total; transposed_slot(u, q, in.x); }
    }
}
#define BA_PAIRS_BLOCK_MAX 256      // blocks: up to here a workgroup per block
#define BA_PAIRS_WAVE_MAX 8192      // blocks: up to here a wavefront per block
// (Round 4, measured at 50 000 keyframes / 700 000 blocks / 97 M pairs and dropped: a wavefront per block for every size -- 4.3 + 7.7 ms to count and fill, the same
// as a thread per block, 3.8 + 8.2 --, and the wavefront form with both lists and q's edge numbers copied into LDS first: 6.4 + 12.0 ms; the divergent serial
// merges are what the time is, not the latency of the list reads.)
// structure scans: one workgroup up to BA_SCAN_ONE_WG entries (a single launch), above that the device-wide three-launch scan (the single workgroup took 3.4 ms
// for the 1.5 M table entries and 1.2 ms for the 700 000 blocks of a 50 000-keyframe map)
#define BA_SCAN_ONE_WG 32768
void ba_launch_pairs_count(const CorbBADev& d, hipStream_t s)
{
    if (d.nu <= BA_PAIRS_BLOCK_MAX) { if (d.nu > 0) hipLaunchKernelGGL(ba_pairs_block_kernel, dim3(d.nu), dim3(256), 0, s, d, 0); }
    else if (d.nu <= BA_PAIRS_WAVE_MAX) hipLaunchKernelGGL(ba_pairs_wave_kernel, dim3((d.nu + 3) / 4), dim3(256), 0, s, d, 0);
    else if (d.row_schur && d.urow) hipLaunchKernelGGL(ba_pairs_row_kernel, dim3(d.nP), dim3(256), 0, s, d, 0);
    else hipLaunchKernelGGL(ba_pairs_count_kernel, dim3((d.nu + 255) / 256), dim3(256), 0, s, d);
    if (d.nu > BA_SCAN_ONE_WG && d.scan_scratch) corb_launch_exclusive_scan(d.pair_off, d.pair_off, (size_t)d.nu, d.scan_scratch, s);
    else hipLaunchKernelGGL(ba_pairs_scan_kernel, dim3(1), dim3(1024), 0, s, d);
}
void ba_launch_pairs_fill(const CorbBADev& d, hipStream_t s)
{
    if (d.nu <= BA_PAIRS_BLOCK_MAX) { if (d.nu > 0) hipLaunchKernelGGL(ba_pairs_block_kernel, dim3(d.nu), dim3(256), 0, s, d, 1); }
    else if (d.nu <= BA_PAIRS_WAVE_MAX) hipLaunchKernelGGL(ba_pairs_wave_kernel, dim3((d.nu + 3) / 4), dim3(256), 0, s, d, 1);
    else if (d.row_schur && d.urow) hipLaunchKernelGGL(ba_pairs_row_kernel, dim3(d.nP), dim3(256), 0, s, d, 1);
    else hipLaunchKernelGGL(ba_pairs_fill_kernel, dim3((d.nu + 255) / 256), dim3(256), 0, s, d);
}

// V_e = W_e C_l with C_l C_l' = Dinv_l = (Hll + lambda I)^-1, i.e. C = L^-T of the Cholesky factor L L' = Hll + lambda I.  Then
//     W_e1 Dinv W_e2' = V_e1 V_e2'
// and the Schur kernel reads ONE array for both operands (with BD = W Dinv next to W it gathered from two 460 MB arrays at 10 000 keyframes and
// was bound by random HBM reads: 1.26 GB fetched per launch).  One thread per (edge, row of the 6 x 3 block); edges with a fixed pose or a fixed
// landmark carry no Schur term.  A landmark whose block is not positive definite (non-finite data) fails the trial like a non-finite Dinv.
__global__ __launch_bounds__(256) void ba_v_kernel(CorbBADev d, double lambda, int* bad, int epoch)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int e = t / 6, r = t - 6 * e;
    if (e >= d.nE) return;
    const int l = d.e_point[e];
    if (l < 0 || d.e_pose[e] < 0) return;
    const double* H = d.Hll + 9 * (size_t)l;
    const double m00 = H[0] + lambda, m10 = H[3], m11 = H[4] + lambda, m20 = H[6], m21 = H[7], m22 = H[8] + lambda;
    // L (lower): l00 l10 l11 l20 l21 l22
    const double l00 = sqrt(m00), i00 = 1.0 / l00;
    const double l10 = m10 * i00, l20 = m20 * i00;
    const double d11 = m11 - l10 * l10, l11 = sqrt(d11), i11 = 1.0 / l11;
    const double l21 = (m21 - l20 * l10) * i11;
    const double d22 = m22 - l20 * l20 - l21 * l21, l22 = sqrt(d22), i22 = 1.0 / l22;
    if (!(m00 > 0) || !(d11 > 0) || !(d22 > 0)) *bad = epoch;
    // V' = L^-1 W' (forward substitution per row of W): v0 = w0 / l00; v1 = (w1 - l10 v0) / l11; v2 = (w2 - l20 v0 - l21 v1) / l22
    const double* W = d.hpl + (size_t)e * 18 + r * 3;
    const double v0 = W[0] * i00, v1 = (W[1] - l10 * v0) * i11, v2 = (W[2] - l20 * v0 - l21 * v1) * i22;
    double* o = d.bd + (size_t)e * 18 + r * 3;
    o[0] = v0; o[1] = v1; o[2] = v2;
}

// ---- row-owner form (round 4) ----
#ifndef BA_ROW_RANGE
#define BA_ROW_RANGE 216          // observations of a keyframe per workgroup: their V blocks (31 104 B) + the four wavefronts' operand scratch (4 x 2 304 B) = 40 320 B: FOUR
                                  // workgroups of four wavefronts per CU -- the 16 wavefronts its 111 VGPRs allow.  Measured at 50 000 keyframes (tools/gpu_row_variants.sh, row kernel +
                                  // combine kernel per trial): 8 wavefronts x 352 observations (two workgroups per CU, the form until late in round 4) 3.41 + 0.36 ms, 4 x 216: 3.03 + 0.41,
                                  // 4 x 220: 3.02 + 0.41, 4 x 208: 3.06 + 0.41, 4 x 192: 3.10 + 0.42, 4 x 128: 3.28 + 0.50, 2 x 128: 3.17 + 0.49, 5 x 224: 4.17, 5 x 288: 4.43; units of 96 / 64
                                  // pairs instead of 128 at 4 x 216: 3.21 + 0.45 / 3.32 + 0.48; forced to 96 VGPRs (five wavefronts per SIMD, 12 bytes of scratch) with five workgroups
                                  // per CU: 4 x 128: 3.21 + 0.50, 4 x 152: 3.93, 4 x 160: 4.42 (ranges of 150-160 observations are slow at either occupancy: 4 x 160 above took 4.26)
#endif
#ifndef BA_ROW_SEG
#define BA_ROW_SEG 128            // pairs per work unit (8 rounds)
#endif
// One wavefront per block (p, q >= p).  The four independent 4x4x4 products of an instruction SPLIT THE CONTRACTION: lane (k = lane>>4, blk =
// (lane>>2)&3, i = lane&3) owns pair 4 blk + k of a group of 16 pairs and feeds row i (then row 4+i) of its BD block and column i (then 4+i) of
// its V block (V = W C, see ba_v_kernel: both operands come from one array); the three landmark axes are three instructions per quadrant of the 6x6 block (padded to 8x8), 12 per group.  Every operand is
// loaded by exactly one lane (with the quadrants as the four blocks every A row was loaded by two lanes and every B column by two: the kernel was
// bound by the texture-address unit, TA_BUSY 73 %).  The four partial sums of a quadrant meet once per block in two exchanges (fixed order).
// XCD-aware: workgroup b runs on XCD b % 8 (round-robin dispatch); XCD x takes the x-th eighth of the block list, i.e. a contiguous range of
// block rows, so the BD blocks of a row and the W blocks of its neighbours are fetched into ONE L2 (in launch order every L2 fetched all of
// them).  Rows / columns 6, 7 of the padded tile only reach outputs that are never stored.
#define BA_SCHUR_WAVES 16       // blocks (wavefronts) per workgroup
// SPLIT > 1: a local window's few blocks with thousands of pairs each -- one workgroup per block, see ba_hpp_mfma_kernel.
template <int SPLIT>
__device__ __forceinline__ void ba_schur_mfma_body(const CorbBADev& d, const double lambda_arg, const int bid, double (*part)[64])
{
    if (d.ctl && d.ctl->stop) return;                        // (a chain of LM iterations that has stopped: see BALMCtl)
    const double lambda = d.ctl ? d.ctl->lambda : lambda_arg;
    const int per = gridDim.x >> 3;                          // (SPLIT == 1: the grid is a multiple of 8 workgroups)
    const int wg = (bid & 7) * per + (bid >> 3);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int u = __builtin_amdgcn_readfirstlane(SPLIT == 1 ? wg * BA_SCHUR_WAVES + wave : bid);       // (wave-uniform: scalar registers, scalar branches)
    const int g0 = SPLIT == 1 ? 0 : 16 * wave;
    constexpr int GS = 16 * SPLIT;
    if (u >= d.nu) return;
    const int4 in = d.uinfo[u];
    const int s = __builtin_amdgcn_readfirstlane(in.x), p = __builtin_amdgcn_readfirstlane(in.y), q = __builtin_amdgcn_readfirstlane(in.z), mir = __builtin_amdgcn_readfirstlane(in.w);
    const int o0 = __builtin_amdgcn_readfirstlane(d.pair_off[u]), n = __builtin_amdgcn_readfirstlane(d.pair_off[u + 1]) - o0;
    const int2* pr = d.pairs + o0;
    // with the row-owner kernel in charge (d.row_schur) this kernel is launched for the keyframes whose V blocks do not fit its LDS, and the first
    // index of a pair is the edge's position in the keyframe's list
    const int* pe = d.pedge + d.poff[p];

    const int k = lane >> 4, blk = (lane >> 2) & 3, i4 = lane & 3;
    const int pl = 4 * blk + k;                              // this lane's pair inside a group of 16
    const int rlo = i4 * 3, rhi = min(4 + i4, 5) * 3;        // rows 6, 7: row 5 again (their products are discarded)
    double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
    if (n > g0) {
        int2 e = pr[min(g0 + pl, n - 1)];
        for (int c0 = g0; c0 < n; c0 += GS) {
            const bool live = c0 + pl < n;                   // past the end of the list the lane re-reads the last pair and feeds A = 0
            // (d.v_kf: the V blocks lie in keyframe-list order -- the first operand at its list position, the second at the position the pair entry names)
            const double* A = d.bd + (size_t)(d.v_kf ? d.poff[p] + e.x : d.row_schur ? pe[e.x] : e.x) * 18; const double* B = d.bd + (size_t)e.y * 18;
            double al[3], ah[3], bl[3], bh[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { al[c] = A[rlo + c]; ah[c] = A[rhi + c]; bl[c] = B[rlo + c]; bh[c] = B[rhi + c]; }
            e = pr[min(c0 + GS + pl, n - 1)];                // the next group's pair travels while this group's operands arrive: one latency per group
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double xl = live ? al[c] : 0.0, xh = live ? ah[c] : 0.0;
                a00 = __builtin_amdgcn_mfma_f64_4x4x4f64(xl, bl[c], a00, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f64_4x4x4f64(xl, bh[c], a01, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f64_4x4x4f64(xh, bl[c], a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f64_4x4x4f64(xh, bh[c], a11, 0, 0, 0);
            }
        }
    }
    // D[blk][i][j] at lane 16 i + 4 blk + j holds the partial sum of the pairs of `blk`: (b0 + b1) + (b2 + b3) on every lane, then lane blk keeps
    // quadrant (blk>>1, blk&1): element row = 4 (blk>>1) + (lane>>4), col = 4 (blk&1) + (lane&3)
    a00 += lx_xor<4>(a00); a01 += lx_xor<4>(a01); a10 += lx_xor<4>(a10); a11 += lx_xor<4>(a11);
    a00 += lx_xor<8>(a00); a01 += lx_xor<8>(a01); a10 += lx_xor<8>(a10); a11 += lx_xor<8>(a11);
    double acc = blk == 0 ? a00 : blk == 1 ? a01 : blk == 2 ? a10 : a11;
    if (SPLIT > 1) {
        part[wave][lane] = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = 0;
#pragma unroll
        for (int w = 0; w < SPLIT; w++) acc += part[w][lane];
    }
    const int row = 4 * (blk >> 1) + k, col = 4 * (blk & 1) + i4;
    if (row >= 6 || col >= 6) return;
    double v = -acc;
    if (p == q) {
        // the reference keeps the upper triangle of a diagonal block (linear_solver_eigen.h:203-232 copies the upper part): mirror it
        if (row > col) return;
        v += d.Hpp[36 * (size_t)p + row * 6 + col] + (row == col ? lambda : 0.0);
        if (d.use_bsr) { double* o = d.bsr_val + (size_t)s * 36; o[row * 6 + col] = v; o[col * 6 + row] = v; }
        else { d.S[(size_t)(6 * p + row) * d.sp + 6 * p + col] = v; d.S[(size_t)(6 * p + col) * d.sp + 6 * p + row] = v; }
        return;
    }
    if (d.use_bsr) { d.bsr_val[(size_t)s * 36 + row * 6 + col] = v; d.bsr_val[(size_t)mir * 36 + col * 6 + row] = v; }
    else { d.S[(size_t)(6 * p + row) * d.sp + 6 * q + col] = v; d.S[(size_t)(6 * q + col) * d.sp + 6 * p + row] = v; }
}
template <int SPLIT>
__global__ __launch_bounds__(SPLIT == 1 ? 64 * BA_SCHUR_WAVES : 64 * SPLIT) void ba_schur_mfma_kernel(CorbBADev d, double lambda)
{
    __shared__ double part[SPLIT == 1 ? 1 : SPLIT][64];
    ba_schur_mfma_body<SPLIT>(d, lambda, (int)blockIdx.x, part);
}
// Local windows (a few keyframes with thousands of observations: both kernels in their SPLIT form, a workgroup per block / per keyframe): the Schur products and
// the reduced right-hand side depend on V only, not on each other -- ONE launch, workgroups [0, nu) take the blocks, [nu, nu + nP) the keyframes.  A call on a
// window is a chain of ~130 dependent launches of 5-15 us each (profiles/r04_lba): every launch saved is ~8 us per LM trial.
__global__ __launch_bounds__(64 * BA_SMALL_SPLIT) void ba_schur_rhs_split_kernel(CorbBADev d, double lambda)
{
    __shared__ double part[BA_SMALL_SPLIT][64];
    __shared__ double part_rhs[16][6];
    if ((int)blockIdx.x < d.nu) ba_schur_mfma_body<BA_SMALL_SPLIT>(d, lambda, (int)blockIdx.x, part);
    else ba_reduced_rhs_lean_body<16>(d, (int)blockIdx.x - d.nu, part_rhs);
}

// Row-owner Schur products (round 4; block_solver.hpp:400-431).  The pair-list kernel above gathers BOTH 144-byte V blocks of every pair in 24-byte
// pieces (24 vector loads per 16 pairs: 28 GB of L2 requests per launch at 27.5 M observations, 8.9 GB of them from HBM, the texture-address unit busy
// for 4.3 ms with the FP64 matrix pipe at 12 %).  Here a workgroup of 16 wavefronts owns block row p:
//   * the V blocks of p's own observations -- the first operand of EVERY pair of the row -- are staged once in LDS (coalesced 16-byte pieces);
//   * a wavefront takes the row's blocks (p, q) one by one; per round of 16 pairs the second operands (16 x 144 B) are fetched by the whole wavefront
//     as 144 sixteen-byte pieces -- three loads per lane instead of twelve -- one round AHEAD into registers, dropped into the wavefront's LDS scratch,
//     and both operands of the matrix instructions are read from LDS;
//   * same lane maps, same accumulation order (pairs ascending in the landmark, four interleaved partial sums met at the end) and therefore the same
//     bits as the pair-list kernel: blocks are written once, no atomics.
// Keyframes with more than BA_ROW_CAP observations of free landmarks stay with the pair-list kernel (d.row_schur makes it skip the others).
__global__ __launch_bounds__(256) void ba_urow_kernel(CorbBADev d)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u > d.nu) return;
    if (u == d.nu) { d.urow[d.nP] = d.nu; return; }
    const int p = d.uinfo[u].y;                             // (every row holds its diagonal block: every urow entry is written)
    if (u == 0 || d.uinfo[u - 1].y != p) d.urow[p] = u;
}
// Work decomposition of the row-owner kernel.  A workgroup owns (keyframe p, RANGE r of its observation list: entries [r BA_ROW_RANGE, (r + 1) BA_ROW_RANGE)) --
// BA_ROW_RANGE V blocks of LDS (216 = 31 KB: FOUR four-wavefront workgroups share a CU; until late in round 4 352 = 50 KB and two eight-wavefront ones), so that one's dependent start-up trips hide behind the others' rounds (one workgroup per CU with the
// whole list in LDS spent 16 k of its 50 k cycles per row on them).  Inside it the work units are the segments of at most BA_ROW_SEG pairs of the blocks' pair
// lists whose first observation lies in the range (lists ascend in it); the blocks of a row are very unequal -- the diagonal block pairs every observation with
// itself (~550 pairs), the next neighbours share 80 / 60 / 45 % of them, the far ones a few dozen -- and with one wavefront per block the diagonal block's 34
// rounds were the length of the row.  A unit's partial block goes to upart[unit][36]; ba_schur_combine_kernel adds a block's units in (range, segment) order.
//   rr_off[p]            first workgroup of keyframe p (exclusive scan of its ranges)
//   wghdr[w]             (first list entry, entries, first unit, end unit)
//   wb_off[w]            first entry of workgroup w in wb_unit;  wb_unit[wb_off[w] + b] = first unit of the row's b-th block in workgroup w (b = nblocks: the end)
//   units[j]             (first pair, pairs, first list entry of the range, -)
__device__ __forceinline__ int ba_pair_lower_bound(const int2* pr, int n, int ia)      // first pair with .x >= ia
{ int a = 0, b = n; while (a < b) { const int mid = (a + b) >> 1; if (pr[mid].x < ia) a = mid + 1; else b = mid; } return a; }
__global__ __launch_bounds__(256) void ba_rr_count_kernel(CorbBADev d)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p > d.nP) return;
    if (p == d.nP) { d.rr_off[p] = 0; d.rowwb[p] = 0; return; }
    const int i0 = d.poff[p], nA = ba_plm_valid(d, i0, d.poff[p + 1]) - i0;
    const int nr = max(1, (nA + BA_ROW_RANGE - 1) / BA_ROW_RANGE);
    d.rr_off[p] = nr; d.rowwb[p] = nr * (d.urow[p + 1] - d.urow[p] + 1);
}
// exclusive scan of a[0 .. n) in place, total into a[n]: one workgroup, a contiguous chunk per thread
__global__ __launch_bounds__(1024) void ba_scan_inplace_kernel(int* a, int n)
{
    __shared__ int part[1024];
    const int t = threadIdx.x, chunk = (n + 1023) / 1024;
    const int i0 = min(n, t * chunk), i1 = min(n, i0 + chunk);
    int sum = 0;
    for (int i = i0; i < i1; i++) sum += a[i];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { const int v = t >= o ? part[t - o] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    int run = part[t] - sum;
    for (int i = i0; i < i1; i++) { const int c = a[i]; a[i] = run; run += c; }
    if (t == 1023) a[n] = part[1023];
}
// One thread per upper block u = (p, b-th block of row p): its pair list ascends in the first observation, so the pairs of range r are the segment between two
// lower bounds.  COUNT: units of (range r, block b) into wb_unit[rowwb[p] + r (nb + 1) + b] (the entry b = nb of every workgroup stays 0); an exclusive scan of
// wb_unit in that order -- workgroup by workgroup, block by block -- turns the counts into first-unit indices, the entry nb into the workgroup's end.  FILL: the units.
// (A thread per ROW walking blocks x ranges was the first form: a hub keyframe with hundreds of blocks would have held the whole structure pass.)
template <bool FILL>
__global__ __launch_bounds__(256) void ba_rr_units_kernel(CorbBADev d)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= d.nu) return;
    const int p = d.uinfo[u].y, u0 = d.urow[p], nb = d.urow[p + 1] - u0, b = u - u0;
    const int nr = d.rr_off[p + 1] - d.rr_off[p];
    const int o0 = d.pair_off[u], n = d.pair_off[u + 1] - o0;
    int lo = 0;
    for (int r = 0; r < nr; r++) {
        const int hi = r + 1 < nr ? ba_pair_lower_bound(d.pairs + o0, n, (r + 1) * BA_ROW_RANGE) : n;
        const int slot = d.rowwb[p] + r * (nb + 1) + b;
        if (!FILL) { d.wb_unit[slot] = (hi - lo + BA_ROW_SEG - 1) / BA_ROW_SEG; if (b == 0) d.wb_unit[slot + nb] = 0; }
        else { int j = d.wb_unit[slot]; for (int k = lo; k < hi; k += BA_ROW_SEG) d.units[j++] = make_int4(o0 + k, min(BA_ROW_SEG, hi - k), r * BA_ROW_RANGE, 0); }
        lo = hi;
    }
}
// header of workgroup w = (keyframe p, range r): (first list entry, entries, first unit, end unit); wb_off[w]
__global__ __launch_bounds__(256) void ba_rr_header_kernel(CorbBADev d)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.nP) return;
    const int i0 = d.poff[p], nA = ba_plm_valid(d, i0, d.poff[p + 1]) - i0;
    const int w0 = d.rr_off[p], nr = d.rr_off[p + 1] - w0, nb = d.urow[p + 1] - d.urow[p];
    for (int r = 0; r < nr; r++) {
        const int wb = d.rowwb[p] + r * (nb + 1);
        d.wb_off[w0 + r] = wb;
        d.wghdr[w0 + r] = make_int4(i0 + r * BA_ROW_RANGE, max(0, min(BA_ROW_RANGE, nA - r * BA_ROW_RANGE)), d.wb_unit[wb], d.wb_unit[wb + nb]);
    }
}
#define ROW_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// The dependent memory round trips of a workgroup are what this kernel costs (one workgroup per CU: nothing else hides them; the first version walked
// header -> binary search -> list -> blocks -> barrier -> block header -> pairs -> operands, ~20 trips = 24 us per row, 4.6 ms per launch).  Now:
// 1 row header (ba_row_header_kernel) -> 2 the thread's list entries + the wavefront's block header -> 3 the row's V pieces + the block's first pairs
// -> 4 the first two rounds' second operands; only then the LDS stores and the one barrier.
#define ROW_NPIECE ((BA_ROW_RANGE * 9 + 64 * BA_ROW_WAVES - 1) / (64 * BA_ROW_WAVES))
__global__ __launch_bounds__(64 * BA_ROW_WAVES) void ba_schur_row_kernel(CorbBADev d)
{
    extern __shared__ double2 row_sm[];                     // [BA_ROW_RANGE][9] the range's V blocks | [BA_ROW_WAVES][16][9] second operands of a round, per wavefront
    const int per = gridDim.x >> 3;                         // XCD x takes the x-th eighth of the workgroups, i.e. of the rows (neighbouring rows share second operands: one L2)
    const int w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (w >= d.n_wg) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#ifdef CORB_DEV
#define ROW_TS(i) do { if (d.row_dbg && lane == 0) d.row_dbg[((size_t)blockIdx.x * BA_ROW_WAVES + wave) * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define ROW_TS(i) do { } while (0)
#endif
    ROW_TS(0);
    const int4 hdr = d.wghdr[w];                            // first list entry, entries, first / end work unit
    const int i0 = hdr.x, nA = hdr.y, j0 = hdr.z, j1 = hdr.w;
    if (j1 <= j0) { if (tid < 6 * BA_ROW_WAVES) d.rpart[(size_t)w * 6 * BA_ROW_WAVES + tid] = 0.0; return; }      // (no observation of a free landmark in the range)
    const double2* bd2 = reinterpret_cast<const double2*>(d.bd);
    const int n9 = nA * 9;
    // The range's share of the reduced right-hand side, sum over its observations of V_e g_l (block_solver.hpp:439-456), comes out of the staged blocks: thread t
    // takes observation t -- its landmark and g_l travel with trips 2 to 4 --, six dot products from LDS after the barrier, a wavefront sum (fixed order), one
    // 6-vector per wavefront to rpart; ba_schur_combine_kernel adds them in (range, wavefront) order.  The separate pass (a wavefront per keyframe gathering
    // 168 bytes per observation: 1.18 ms per trial at 27.5 M observations) is gone.
    static_assert(BA_ROW_RANGE <= 64 * BA_ROW_WAVES, "an observation per thread");
    const int my_edge = tid < nA ? d.pedge[i0 + tid] : -1;
    // ---- trip 2: list entries of this thread's pieces, the wavefront's first two work units ----
    // (a wavefront's 64 pieces are consecutive; past the end of the range's pieces the lanes of its last wavefront repeat the last piece)
    int pe[ROW_NPIECE];
#pragma unroll
    for (int j = 0; j < ROW_NPIECE; j++) { const int mb = 64 * BA_ROW_WAVES * j + 64 * wave; pe[j] = mb < n9 ? d.pedge[i0 + min(mb + lane, n9 - 1) / 9] : 0; }
    int ju = j0 + wave;                                     // this wavefront's units: ju, ju + 8, ...
    int n = 0, n2 = 0, base = 0, base2 = 0; const int2* pr = d.pairs; const int2* pr2 = d.pairs;
    if (ju < j1) { const int4 un = d.units[ju]; n = __builtin_amdgcn_readfirstlane(un.y); pr = d.pairs + __builtin_amdgcn_readfirstlane(un.x); base = __builtin_amdgcn_readfirstlane(un.z); }
    if (ju + BA_ROW_WAVES < j1) { const int4 un = d.units[ju + BA_ROW_WAVES]; n2 = __builtin_amdgcn_readfirstlane(un.y); pr2 = d.pairs + __builtin_amdgcn_readfirstlane(un.x); base2 = __builtin_amdgcn_readfirstlane(un.z); }
    // ---- trip 3: the range's V pieces, straight into LDS (global_load_lds_dwordx4: destination = the wavefront's base + 16 lane, no staging registers --
    // eight pieces per thread held in registers next to the operand sets below spilled 4 GB of scratch per launch), the pairs of the first two units ----
#pragma unroll
    for (int j = 0; j < ROW_NPIECE; j++) {
        const int mb = 64 * BA_ROW_WAVES * j + 64 * wave;    // wave-uniform
        if (mb < n9) {
            const int m = min(mb + lane, n9 - 1);
#ifdef CORB_DEV
            const size_t src_ = (d.row_abl & 16) ? ((size_t)i0 * 9 + m) % ((size_t)d.nfree_edges * 9) : (size_t)pe[j] * 9 + (m - 9 * (m / 9));      // 16: the pieces from CONSECUTIVE addresses (timing experiment)
#else
            const size_t src_ = (size_t)pe[j] * 9 + (m - 9 * (m / 9));
#endif
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bd2 + src_),
                                             (__attribute__((address_space(3))) void*)(row_sm + mb + lane), 16, 0, 0);
        }
    }
    const int k = lane >> 4, blk = (lane >> 2) & 3, i4 = lane & 3;
    const int pl = 4 * blk + k;                              // this lane's pair inside a round of 16 (lane maps: see ba_schur_mfma_kernel)
    const int rlo = i4 * 3, rhi = min(4 + i4, 5) * 3;
    // the three pieces this lane fetches of a round's 144: piece m = lane + 64 j -> pair m / 9, part m % 9
    const int m1 = lane + 64, m2 = min(lane + 128, 143);
    const int pa0 = lane / 9, pa1 = m1 / 9, pa2 = m2 / 9;
    const int pt0 = lane - 9 * pa0, pt1 = m1 - 9 * pa1, pt2 = m2 - 9 * pa2;
    // a unit has at most BA_ROW_SEG = 128 pairs: two batches of 64 pair entries (lane j holds pair j of its batch), fetched whole up front
    int2 entC = make_int2(0, 0), entN = make_int2(0, 0), ent2C = make_int2(0, 0), ent2N = make_int2(0, 0);
    const int my_lm = my_edge >= 0 ? d.e_point[my_edge] : -1;
    if (n > 0) { entC = pr[min(lane, n - 1)]; entN = pr[min(64 + lane, n - 1)]; }
    if (n2 > 0) { ent2C = pr2[min(lane, n2 - 1)]; ent2N = pr2[min(64 + lane, n2 - 1)]; }
    double2 bx0, bx1, bx2, by0, by1, by2;                    // second operands of the next two rounds, in two register sets that alternate
#ifdef CORB_DEV
    const int abl = d.row_abl;       // timing experiments (results wrong): 1 = second operands always from one address, 2 = no scratch stores, 4 = no matrix instructions, 8 = no operand reads
#define ROW_LOADB(r0, r1, r2, ent, off) do { const int e0_ = (abl & 1) ? 0 : __shfl((ent).y, (off) + pa0), e1_ = (abl & 1) ? 0 : __shfl((ent).y, (off) + pa1), e2_ = (abl & 1) ? 0 : __shfl((ent).y, (off) + pa2); \
        r0 = bd2[(size_t)e0_ * 9 + pt0]; r1 = bd2[(size_t)e1_ * 9 + pt1]; r2 = bd2[(size_t)e2_ * 9 + pt2]; } while (0)
#else
    const int abl = 0;
#define ROW_LOADB(r0, r1, r2, ent, off) do { const int e0_ = __shfl((ent).y, (off) + pa0), e1_ = __shfl((ent).y, (off) + pa1), e2_ = __shfl((ent).y, (off) + pa2); \
        r0 = bd2[(size_t)e0_ * 9 + pt0]; r1 = bd2[(size_t)e1_ * 9 + pt1]; r2 = bd2[(size_t)e2_ * 9 + pt2]; } while (0)
#endif
    ROW_TS(1);
    // ---- trip 4: the second operands of the first unit's first two rounds are in flight when the barrier is reached ----
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
    if (my_lm >= 0) { const double* g = d.db + 3 * (size_t)my_lm; g0 = g[0]; g1 = g[1]; g2 = g[2]; }
    if (n > 0) { ROW_LOADB(bx0, bx1, bx2, entC, 0); ROW_LOADB(by0, by1, by2, entC, 16); }
    ROW_TS(2);
    __syncthreads();                                        // (carries the vmcnt(0) that lands the LDS-direct loads)
    ROW_TS(3);
    double2* scr = row_sm + (size_t)BA_ROW_RANGE * 9 + wave * 144;
    const double* Asm = reinterpret_cast<const double*>(row_sm);
    const double* Bsm = reinterpret_cast<const double*>(scr);
    if (64 * wave < nA) {                                   // (wave-uniform) the wavefront's observations: V_e g_l, summed over the lanes
        const double* Vt = Asm + (size_t)min(tid, nA - 1) * 18;
        double rv[6];
#pragma unroll
        for (int a = 0; a < 6; a++) rv[a] = Vt[3 * a] * g0 + Vt[3 * a + 1] * g1 + Vt[3 * a + 2] * g2;      // (g = 0 past the range and for a fixed landmark)
#pragma unroll
        for (int a = 0; a < 6; a++) {
            rv[a] = lx_wave_sum(rv[a]);
        }
        if (lane < 6) d.rpart[((size_t)w * BA_ROW_WAVES + wave) * 6 + lane] = lane == 0 ? rv[0] : lane == 1 ? rv[1] : lane == 2 ? rv[2] : lane == 3 ? rv[3] : lane == 4 ? rv[4] : rv[5];
    } else if (lane < 6) d.rpart[((size_t)w * BA_ROW_WAVES + wave) * 6 + lane] = 0.0;
    for (int turn = 0; ju < j1; turn++, ju += BA_ROW_WAVES) {
        double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
        if (n > 0) {
            // one round of 16 pairs: the set's registers -> the wavefront's scratch, PREFETCH (a statement), operands from LDS, 12 matrix instructions.
            // Software pipeline (4 wavefronts per SIMD cannot hide an L2 / HBM round trip by themselves): the second operands travel TWO rounds ahead in two
            // register sets that alternate -- no register of an outstanding load is moved or read before its turn; past the end of the list the entries repeat
            // the last pair (valid addresses, A = 0).
#define ROW_ROUND(ENT, t, w0, w1, w2, PREFETCH) do { \
                const int ia_ = __shfl((ENT).x, 16 * (t) + pl) - base; \
                const bool live_ = c0 + 16 * (t) + pl < n;       /* past the end of the list: the last pair again, with A = 0 */ \
                ROW_WAVE_SYNC();                                 /* (the previous round's operand reads are issued: LDS serves a wavefront in order) */ \
                if (!(abl & 2)) { scr[lane] = w0; scr[m1] = w1; if (lane < 16) scr[m2] = w2; } else { a00 += w0.x + w1.x + w2.x; } \
                PREFETCH; \
                ROW_WAVE_SYNC(); \
                const double* A_ = Asm + (size_t)ia_ * 18; const double* B_ = Bsm + pl * 18; \
                double al_[3], ah_[3], bl_[3], bh_[3]; \
                if (!(abl & 8)) { _Pragma("unroll") for (int c = 0; c < 3; c++) { al_[c] = A_[rlo + c]; ah_[c] = A_[rhi + c]; bl_[c] = B_[rlo + c]; bh_[c] = B_[rhi + c]; } } \
                else { _Pragma("unroll") for (int c = 0; c < 3; c++) { al_[c] = ah_[c] = (double)ia_; bl_[c] = bh_[c] = (double)pl; } } \
                _Pragma("unroll") for (int c = 0; c < 3; c++) { \
                    const double xl_ = live_ ? al_[c] : 0.0, xh_ = live_ ? ah_[c] : 0.0; \
                    if (!(abl & 4)) { \
                    a00 = __builtin_amdgcn_mfma_f64_4x4x4f64(xl_, bl_[c], a00, 0, 0, 0); \
                    a01 = __builtin_amdgcn_mfma_f64_4x4x4f64(xl_, bh_[c], a01, 0, 0, 0); \
                    a10 = __builtin_amdgcn_mfma_f64_4x4x4f64(xh_, bl_[c], a10, 0, 0, 0); \
                    a11 = __builtin_amdgcn_mfma_f64_4x4x4f64(xh_, bh_[c], a11, 0, 0, 0); \
                    } else { a00 += xl_ * bl_[c]; a01 += xl_ * bh_[c]; a10 += xh_ * bl_[c]; a11 += xh_ * bh_[c]; } \
                } } while (0)
            // a full first batch (more than 48 pairs): straight-line code, every load unconditional -- a load or a round under a condition makes the
            // compiler's vmcnt bookkeeping fall back to waiting for EVERYTHING in flight, the latency this pipeline hides
            int c0 = 0;
            if (n > 48) {
                ROW_ROUND(entC, 0, bx0, bx1, bx2, ROW_LOADB(bx0, bx1, bx2, entC, 32));
                ROW_ROUND(entC, 1, by0, by1, by2, ROW_LOADB(by0, by1, by2, entC, 48));
                ROW_ROUND(entC, 2, bx0, bx1, bx2, ROW_LOADB(bx0, bx1, bx2, entN, 0));
                ROW_ROUND(entC, 3, by0, by1, by2, ROW_LOADB(by0, by1, by2, entN, 16));
                c0 = 64; entC = entN;
                if (n > 64 + 48) {                                // a full second batch: its last two rounds prefetch nothing new
                    ROW_ROUND(entC, 0, bx0, bx1, bx2, ROW_LOADB(bx0, bx1, bx2, entC, 32));
                    ROW_ROUND(entC, 1, by0, by1, by2, ROW_LOADB(by0, by1, by2, entC, 48));
                    ROW_ROUND(entC, 2, bx0, bx1, bx2, (void)0);
                    ROW_ROUND(entC, 3, by0, by1, by2, (void)0);
                    c0 = 128;
                }
            }
            // the last batch of one to three rounds (0 .. 48 pairs left): its first two rounds' operands are in flight already
            const int nr = n > c0 ? (n - c0 + 15) >> 4 : 0;     // wave-uniform
            if (nr > 0) {
                if (nr > 2) ROW_ROUND(entC, 0, bx0, bx1, bx2, ROW_LOADB(bx0, bx1, bx2, entC, 32)); else ROW_ROUND(entC, 0, bx0, bx1, bx2, (void)0);
                if (nr > 1) {
                    ROW_ROUND(entC, 1, by0, by1, by2, (void)0);
                    if (nr > 2) ROW_ROUND(entC, 2, bx0, bx1, bx2, (void)0);
                }
            }
#undef ROW_ROUND
        }
        // D[blk][i][j] at lane 16 i + 4 blk + j holds the partial sum of the pairs of `blk`: (b0 + b1) + (b2 + b3) on every lane, then lane blk keeps
        // quadrant (blk>>1, blk&1): the unit's partial block, row-major
        a00 += lx_xor<4>(a00); a01 += lx_xor<4>(a01); a10 += lx_xor<4>(a10); a11 += lx_xor<4>(a11);
        a00 += lx_xor<8>(a00); a01 += lx_xor<8>(a01); a10 += lx_xor<8>(a10); a11 += lx_xor<8>(a11);
        { const int row_ = 4 * (blk >> 1) + k, col_ = 4 * (blk & 1) + i4;
          if (row_ < 6 && col_ < 6) d.upart[(size_t)ju * 36 + row_ * 6 + col_] = blk == 0 ? a00 : blk == 1 ? a01 : blk == 2 ? a10 : a11; }
        ROW_TS(turn == 0 ? 4 : 5);
#ifdef CORB_DEV
        if (d.row_dbg && lane == 0 && turn == 0) d.row_dbg[((size_t)blockIdx.x * BA_ROW_WAVES + wave) * 8 + 7] = n;
#endif
        // the next unit: the second one's pairs are here already (fetched with the first one's), a third one's are fetched now (its trips are not hidden)
        if (turn == 0) { n = n2; base = base2; entC = ent2C; entN = ent2N; }
        else {
            n = 0;
            if (ju + BA_ROW_WAVES < j1) { const int4 un = d.units[ju + BA_ROW_WAVES]; n = __builtin_amdgcn_readfirstlane(un.y); pr = d.pairs + __builtin_amdgcn_readfirstlane(un.x); base = __builtin_amdgcn_readfirstlane(un.z); }
            if (n > 0) { entC = pr[min(lane, n - 1)]; entN = pr[min(64 + lane, n - 1)]; }
        }
        if (n > 0 && ju + BA_ROW_WAVES < j1) { ROW_LOADB(bx0, bx1, bx2, entC, 0); ROW_LOADB(by0, by1, by2, entC, 16); }
    }
#undef ROW_LOADB
}
// ---- Round 6: the same products as ONE STREAM OF ROUNDS per wavefront, second operands straight into the matrix instructions' registers ----
// ba_schur_row_kernel above pays, per work unit, the dependent trips unit header -> pair entries -> second operands with nothing else in flight (~3 k of a later
// unit's 6.7 k cycles; a wavefront has 4-7 units of 2-8 rounds), and moves every second operand global -> registers -> LDS scratch -> registers (the LDS pipe of the CU
// is what its rounds wait for: 95 of a round's ~97 cycles per CU).  Here
//   * the structure pass files every wavefront's rounds as ONE padded stream (ba_rr_stream_kernel): 16 entries per round, (position of the first operand in the
//     range's LDS block, as a byte offset | LAST, edge of the second operand) -- the pad entries name a block of zeros behind the range's blocks, LAST marks a unit's last
//     round; a wavefront's units follow each other in the order of their partial blocks (unit j0 + wave, + BA_ROW_WAVES, ...), so the kernel counts its flushes;
//     a wavefront's stream is padded with dead rounds to a multiple of ROW_NSET rounds;
//   * the loop below runs over the stream ROW_NSET rounds per trip, straight-line (a round or a load under a condition makes the compiler's vmcnt bookkeeping wait
//     for everything in flight): the second operands travel ROW_NSET rounds ahead -- across unit boundaries, nothing to set up per unit -- in ROW_NSET register
//     sets, the entries a group of ROW_NSET rounds ahead of them;
//   * lane (k, blk, i4) of the matrix instruction takes rows i4 and min(4 + i4, 5) of ITS pair's V block from global memory itself: two 24-byte pieces per lane, four
//     load instructions per round instead of three, no scratch, no wave barrier, 9 KB less LDS per workgroup; the first operands stay in LDS as before.
// Same lane maps and the same accumulation order as ba_schur_row_kernel (pairs ascending, four interleaved partial sums met at the end): the same bits.
#define ROW_DEAD 0x40000000
#define ROW_LAST 0x20000000
#ifndef ROW_NSET
#define ROW_NSET 2
#endif
#ifndef ROW_ABL
#define ROW_ABL 0           // timing experiments (results wrong): 1 = every second operand from block 0 / 1, 2 = no first-operand staging, 4 = no matrix instructions, 8 = stop behind the barrier
#endif
// Which wavefront takes which units of a workgroup: longest unit first, each to the wavefront with the fewest rounds so far (units dealt round-robin left the
// wavefronts of a workgroup 20-30 % apart -- the diagonal block's units are 7-8 rounds, the far blocks' 1-2 -- and the workgroup's LDS waits for the slowest).
// One thread per workgroup: wunit[j0 .. j1) = its units grouped by wavefront in the order they are worked on, wave_ucnt[w] = units per wavefront,
// wave_off[w BA_ROW_WAVES + v] = the wavefront's rounds, padded to a multiple of ROW_NSET (exclusive scan by the caller, wave_off[n] = the total).
#define ROW_LPT_MAX 96
__global__ __launch_bounds__(256) void ba_rr_assign_kernel(CorbBADev d)
{
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w > d.n_wg) return;
    if (w == d.n_wg) { d.wave_off[(size_t)w * BA_ROW_WAVES] = 0; return; }
    const int4 hdr = d.wghdr[w];
    const int j0 = hdr.z, nu = hdr.w - hdr.z;
    int load[BA_ROW_WAVES], cnt[BA_ROW_WAVES];
#pragma unroll
    for (int v = 0; v < BA_ROW_WAVES; v++) { load[v] = 0; cnt[v] = 0; }
    if (nu > 0 && nu <= ROW_LPT_MAX) {
        unsigned short ord[ROW_LPT_MAX]; unsigned char nr[ROW_LPT_MAX], wv[ROW_LPT_MAX];
        for (int k = 0; k < nu; k++) { nr[k] = (unsigned char)((d.units[j0 + k].y + 15) >> 4); ord[k] = (unsigned short)k; }
        for (int k = 1; k < nu; k++) {                          // stable insertion sort, longest first (ties: the lower unit first)
            const unsigned short o = ord[k]; int m = k;
            while (m > 0 && nr[ord[m - 1]] < nr[o]) { ord[m] = ord[m - 1]; m--; }
            ord[m] = o;
        }
        for (int k = 0; k < nu; k++) {
            int best = 0;
#pragma unroll
            for (int v = 1; v < BA_ROW_WAVES; v++) if (load[v] < load[best]) best = v;
            wv[ord[k]] = (unsigned char)best; load[best] += nr[ord[k]]; cnt[best]++;
        }
        int pos[BA_ROW_WAVES]; pos[0] = 0;
#pragma unroll
        for (int v = 1; v < BA_ROW_WAVES; v++) pos[v] = pos[v - 1] + cnt[v - 1];
        for (int k = 0; k < nu; k++) { const int u = ord[k]; d.wunit[j0 + pos[wv[u]]++] = j0 + u; }      // (per wavefront: longest first)
    } else {
        // a hub keyframe's workgroup (more units than the table holds): round-robin, in two passes over the units
        for (int k = 0; k < nu; k++) { load[k % BA_ROW_WAVES] += (d.units[j0 + k].y + 15) >> 4; cnt[k % BA_ROW_WAVES]++; }
        int pos = 0;
        for (int v = 0; v < BA_ROW_WAVES; v++) for (int k = v; k < nu; k += BA_ROW_WAVES) d.wunit[j0 + pos++] = j0 + k;
    }
    static_assert(BA_ROW_WAVES == 4, "wave_ucnt packs four counts");
    d.wave_ucnt[w] = make_int4(cnt[0], cnt[1], cnt[2], cnt[3]);
#pragma unroll
    for (int v = 0; v < BA_ROW_WAVES; v++) d.wave_off[(size_t)w * BA_ROW_WAVES + v] = (load[v] + ROW_NSET - 1) / ROW_NSET * ROW_NSET;
}
// the stream of wavefront (w, v), written by one wavefront: 64 entries (four rounds) per step, coalesced
__global__ __launch_bounds__(256) void ba_rr_stream_kernel(CorbBADev d)
{
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= d.n_wg * BA_ROW_WAVES) return;
    const int w = t / BA_ROW_WAVES, v = t - w * BA_ROW_WAVES;
    const int4 hdr = d.wghdr[w], uc = d.wave_ucnt[w];
    const int first = hdr.z + (v > 0 ? uc.x : 0) + (v > 1 ? uc.y : 0) + (v > 2 ? uc.z : 0), n_units = v == 0 ? uc.x : v == 1 ? uc.y : v == 2 ? uc.z : uc.w;
    int2* out = d.row_stream + (size_t)d.wave_off[t] * 16;
    const int total = (d.wave_off[t + 1] - d.wave_off[t]) * 16;
    int filled = 0; int2 lastp = make_int2(0, 0);
    for (int k = 0; k < n_units; k++) {
        const int4 un = d.units[d.wunit[first + k]];
        const int np = un.y, ne = ((np + 15) >> 4) * 16;
        for (int i = lane; i < ne; i += 64) {
            const int2 pe = d.pairs[un.x + min(i, np - 1)];
            out[filled + i] = make_int2(((i < np ? pe.x - un.z : BA_ROW_RANGE) * 144) | (i >= ne - 16 ? ROW_LAST : 0), pe.y);
        }
        lastp = d.pairs[un.x + np - 1];
        filled += ne;
    }
    for (int i = filled + lane; i < total; i += 64) out[i] = make_int2(BA_ROW_RANGE * 144, lastp.y);      // dead rounds: the zero block, a valid edge, no flush
}
#ifdef ROW_WPE
__attribute__((amdgpu_waves_per_eu(ROW_WPE, ROW_WPE)))
#endif
__global__ __launch_bounds__(64 * BA_ROW_WAVES) void ba_schur_row_stream_kernel(CorbBADev d)
{
    extern __shared__ double2 row_sm[];                     // [BA_ROW_RANGE + 1][9] the range's V blocks, then a block of zeros (the first operand of the pad entries)
    const int per = gridDim.x >> 3;
    const int w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (w >= d.n_wg) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int4 hdr = d.wghdr[w];
    const int i0 = hdr.x, nA = hdr.y, j0 = hdr.z, j1 = hdr.w;
    if (j1 <= j0) { if (tid < 6 * BA_ROW_WAVES) d.rpart[(size_t)w * 6 * BA_ROW_WAVES + tid] = 0.0; return; }
    const int wv_ = __builtin_amdgcn_readfirstlane(w * BA_ROW_WAVES + wave);
    const int4 uc_ = d.wave_ucnt[w];
    // this wavefront's units in the order of its stream (ba_rr_assign_kernel): the id of the next one is fetched a flush ahead
    const int* ulist = d.wunit + j0 + (wave > 0 ? uc_.x : 0) + (wave > 1 ? uc_.y : 0) + (wave > 2 ? uc_.z : 0);
    const int ulast = max((wave == 0 ? uc_.x : wave == 1 ? uc_.y : wave == 2 ? uc_.z : uc_.w) - 1, 0);
    const int ro0 = __builtin_amdgcn_readfirstlane(d.wave_off[wv_]), ngrp = (__builtin_amdgcn_readfirstlane(d.wave_off[wv_ + 1]) - ro0) / ROW_NSET;       // (one batch with the header)
    const double2* bd2 = reinterpret_cast<const double2*>(d.bd);
    const int n9 = nA * 9;
    const int my_edge = tid < nA ? d.pedge[i0 + tid] : -1;
    int pe[ROW_NPIECE];
#pragma unroll
    for (int j = 0; j < ROW_NPIECE; j++) { const int mb = 64 * BA_ROW_WAVES * j + 64 * wave; pe[j] = (mb < n9 && !d.v_kf) ? d.pedge[i0 + min(mb + lane, n9 - 1) / 9] : 0; }
    const int k = lane >> 4, blk = (lane >> 2) & 3, i4 = lane & 3;
    const int pl = 4 * blk + k;                            // this lane's pair inside a round of 16 (lane maps: see ba_schur_mfma_kernel)
    const int rlo = i4 * 3, rhi = min(4 + i4, 5) * 3;
    // the stream: a lane reads ITS pair's entry of a round itself (the four lanes of a pair the same 8 bytes) -- no cross-lane traffic in the loop; the entries of a group of
    // ROW_NSET rounds travel three groups ahead of the group that is multiplied (one group ahead of the second operands they name)
    const int2* st = d.row_stream + (size_t)ro0 * 16 + pl;
    const int last_r = max(ngrp * ROW_NSET - 1, 0);
    int2 eA[ROW_NSET], eB[ROW_NSET], eC[ROW_NSET];
#pragma unroll
    for (int t = 0; t < ROW_NSET; t++) { eA[t] = make_int2(BA_ROW_RANGE * 144, 0); eB[t] = eA[t]; eC[t] = eA[t]; }
    if (ngrp > 0) {
#pragma unroll
        for (int t = 0; t < ROW_NSET; t++) { eA[t] = st[(size_t)min(t, last_r) * 16]; eB[t] = st[(size_t)min(ROW_NSET + t, last_r) * 16]; eC[t] = st[(size_t)min(2 * ROW_NSET + t, last_r) * 16]; }
    }
#pragma unroll
    for (int j = 0; j < ROW_NPIECE; j++) {
        const int mb = 64 * BA_ROW_WAVES * j + 64 * wave;
        if (mb + lane < n9 && !(ROW_ABL & 2)) {               // (per lane: a piece past the range's end would land in the block of zeros behind it)
            const int m = mb + lane;
            const size_t src_ = d.v_kf ? (size_t)i0 * 9 + m : (size_t)pe[j] * 9 + (m - 9 * (m / 9));      // keyframe-list order: the range's blocks are one contiguous stretch
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bd2 + src_),
                                             (__attribute__((address_space(3))) void*)(row_sm + mb + lane), 16, 0, 0);
        }
    }
    if (tid < 9) row_sm[BA_ROW_RANGE * 9 + tid] = make_double2(0.0, 0.0);
    const int my_lm = my_edge >= 0 ? d.e_point[my_edge] : -1;
    // second operands of a round: rows i4 and min(4 + i4, 5) of the pair's block, 24 bytes each
    double bl[ROW_NSET][3], bh[ROW_NSET][3];
#define STREAM_LOADB(set, e_in) do { const int e_ = (ROW_ABL & 1) ? (pl & 1) : (e_in); const double* v_ = d.bd + (size_t)e_ * 18; \
        _Pragma("unroll") for (int c = 0; c < 3; c++) { bl[set][c] = v_[rlo + c]; bh[set][c] = v_[rhi + c]; } } while (0)
    static_assert(ROW_NSET >= 2 && ROW_NSET <= 4, "prefetch depth");
    if (ngrp > 0) {
#pragma unroll
        for (int s_ = 0; s_ < ROW_NSET; s_++) STREAM_LOADB(s_, eA[s_].y);
    }
    __syncthreads();                                        // (carries the vmcnt(0) that lands the LDS-direct loads)
    if (ROW_ABL & 8) { if (tid == 0) d.rpart[(size_t)w * 6 * BA_ROW_WAVES] = bl[0][0] + row_sm[0].x; return; }      // (timing experiment: the workgroup's start-up alone)
    const double* Asm = reinterpret_cast<const double*>(row_sm);
    // (the range's share of the reduced right-hand side follows the products: g_l of the thread's observation -- a FOURTH dependent trip behind header, list entry and
    //  landmark -- is requested here, behind the barrier, and used after the loop, instead of lengthening the chain of trips in front of the barrier)
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
    if (my_lm >= 0) { const double* gp = d.db + 3 * (size_t)my_lm; g0 = gp[0]; g1 = gp[1]; g2 = gp[2]; }
    double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
    int uk = 0, ju = __builtin_amdgcn_readfirstlane(ulist[0]), ju_next = ulist[min(1, ulast)];      // (ju_next stays a vector register until its flush: reading it
                                                                                                      //  into a scalar right after the load would wait for EVERY load in flight)
    const char* Ab = reinterpret_cast<const char*>(row_sm);
    // one group of ROW_NSET rounds: this group's entries `ec`, the next group's `en` (the sets' next values)
#define STREAM_GROUP(ec, en) do { \
        _Pragma("unroll") for (int t = 0; t < ROW_NSET; t++) { \
            const int x_ = (ec)[t].x; \
            const bool last_ = __builtin_amdgcn_readfirstlane(x_) & ROW_LAST; \
            const double* A_ = reinterpret_cast<const double*>(Ab + (x_ & 0xFFFF)); \
            double al_[3], ah_[3]; \
            _Pragma("unroll") for (int c = 0; c < 3; c++) { al_[c] = A_[rlo + c]; ah_[c] = A_[rhi + c]; } \
            _Pragma("unroll") for (int c = 0; c < 3; c++) { \
                if (ROW_ABL & 4) { a00 += al_[c] * bl[t][c]; a01 += al_[c] * bh[t][c]; a10 += ah_[c] * bl[t][c]; a11 += ah_[c] * bh[t][c]; } else { \
                a00 = __builtin_amdgcn_mfma_f64_4x4x4f64(al_[c], bl[t][c], a00, 0, 0, 0); \
                a01 = __builtin_amdgcn_mfma_f64_4x4x4f64(al_[c], bh[t][c], a01, 0, 0, 0); \
                a10 = __builtin_amdgcn_mfma_f64_4x4x4f64(ah_[c], bl[t][c], a10, 0, 0, 0); \
                a11 = __builtin_amdgcn_mfma_f64_4x4x4f64(ah_[c], bh[t][c], a11, 0, 0, 0); } \
            } \
            STREAM_LOADB(t, (en)[t].y);                         /* the set is free: the same round of the next group */ \
            if (last_) {                                        /* (wave-uniform) the unit's partial block, row-major: see ba_schur_row_kernel */ \
                a00 += lx_xor<4>(a00); a01 += lx_xor<4>(a01); a10 += lx_xor<4>(a10); a11 += lx_xor<4>(a11); \
                a00 += lx_xor<8>(a00); a01 += lx_xor<8>(a01); a10 += lx_xor<8>(a10); a11 += lx_xor<8>(a11); \
                const int row_ = 4 * (blk >> 1) + k, col_ = 4 * (blk & 1) + i4; \
                if (row_ < 6 && col_ < 6) d.upart[(size_t)ju * 36 + row_ * 6 + col_] = blk == 0 ? a00 : blk == 1 ? a01 : blk == 2 ? a10 : a11; \
                uk++; ju = __builtin_amdgcn_readfirstlane(ju_next); ju_next = ulist[min(uk + 1, ulast)]; a00 = a01 = a10 = a11 = 0; \
            } \
        } } while (0)
    // TWO groups per trip: a set's loop-carried value (loaded in the trip's second half) and the value loaded in the first half then have disjoint lifetimes and share
    // their registers; with one group per trip the compiler loaded into fresh registers and MOVED them into the loop-carried ones at the end of the trip -- behind a
    // vmcnt(0) that drained every load in flight.  (The entry registers ARE moved at the end of a group: they were loaded two groups before, the wait is for them only.)
    int g = 0;
    for (; g + 2 <= ngrp; g += 2) {
        int2 eD[ROW_NSET], eE[ROW_NSET];
#pragma unroll
        for (int t = 0; t < ROW_NSET; t++) eD[t] = st[(size_t)min((g + 3) * ROW_NSET + t, last_r) * 16];     // (unconditional: past the end, the last round again)
        STREAM_GROUP(eA, eB);
#pragma unroll
        for (int t = 0; t < ROW_NSET; t++) eE[t] = st[(size_t)min((g + 4) * ROW_NSET + t, last_r) * 16];
        STREAM_GROUP(eB, eC);
#pragma unroll
        for (int t = 0; t < ROW_NSET; t++) { eA[t] = eC[t]; eB[t] = eD[t]; eC[t] = eE[t]; }
    }
    if (g < ngrp) STREAM_GROUP(eA, eB);
#undef STREAM_GROUP
#undef STREAM_LOADB
    if (64 * wave < nA) {
        const double* Vt = Asm + (size_t)min(tid, nA - 1) * 18;
        double rv[6];
#pragma unroll
        for (int a = 0; a < 6; a++) rv[a] = Vt[3 * a] * g0 + Vt[3 * a + 1] * g1 + Vt[3 * a + 2] * g2;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            rv[a] = lx_wave_sum(rv[a]);
        }
        if (lane < 6) d.rpart[((size_t)w * BA_ROW_WAVES + wave) * 6 + lane] = lane == 0 ? rv[0] : lane == 1 ? rv[1] : lane == 2 ? rv[2] : lane == 3 ? rv[3] : lane == 4 ? rv[4] : rv[5];
    } else if (lane < 6) d.rpart[((size_t)w * BA_ROW_WAVES + wave) * 6 + lane] = 0.0;
}
// S(p, q) = [p == q] (Hpp + lambda I) - the block's units, added in (range, segment) order; blocks are written once (and mirrored), no atomics
// The workgroups past nblk_blocks: b_schur = b_p - the keyframe's rpart vectors, added in (range, wavefront) order.
__global__ __launch_bounds__(256) void ba_schur_combine_kernel(CorbBADev d, double lambda, int nblk_blocks)
{
    if ((int)blockIdx.x >= nblk_blocks) {
        const int tr = ((int)blockIdx.x - nblk_blocks) * 256 + threadIdx.x, p = tr / 6, a = tr - 6 * p;
        if (p >= d.nP) return;
        double v = 0;
        for (size_t k = (size_t)d.rr_off[p] * BA_ROW_WAVES; k < (size_t)d.rr_off[p + 1] * BA_ROW_WAVES; k++) v += d.rpart[k * 6 + a];
        d.x[6 * (size_t)p + a] = d.b[6 * (size_t)p + a] - v;
        return;
    }
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int u = t / 36, e = t - 36 * u;
    if (u >= d.nu) return;
    const int4 in = d.uinfo[u];
    const int s = in.x, p = in.y, q = in.z, mir = in.w;
    const int b = u - d.urow[p], w0 = d.rr_off[p], nr = d.rr_off[p + 1] - w0;
    double acc = 0;
    for (int r = 0; r < nr; r++) {
        const int wb = d.wb_off[w0 + r];
        for (int j = d.wb_unit[wb + b], je = d.wb_unit[wb + b + 1]; j < je; j++) acc += d.upart[(size_t)j * 36 + e];
    }
    const int row = e / 6, col = e - 6 * row;
    double v = -acc;
    if (p == q) {
        if (row > col) return;                                // the reference keeps the upper triangle of a diagonal block (linear_solver_eigen.h:203-232): mirror it
        v += d.Hpp[36 * (size_t)p + e] + (row == col ? lambda : 0.0);
        if (d.use_bsr) { double* o = d.bsr_val + (size_t)s * 36; o[row * 6 + col] = v; o[col * 6 + row] = v; }
        else { d.S[(size_t)(6 * p + row) * d.sp + 6 * p + col] = v; d.S[(size_t)(6 * p + col) * d.sp + 6 * p + row] = v; }
        return;
    }
    if (d.use_bsr) { d.bsr_val[(size_t)s * 36 + e] = v; d.bsr_val[(size_t)mir * 36 + col * 6 + row] = v; }
    else { d.S[(size_t)(6 * p + row) * d.sp + 6 * q + col] = v; d.S[(size_t)(6 * q + col) * d.sp + 6 * p + row] = v; }
}
#define BA_ROW_LDS ((size_t)(BA_ROW_RANGE * 144 + BA_ROW_WAVES * 16 * 144))

void ba_schur_mfma_launch(const CorbBADev& d, double lambda, int* bad, int epoch, hipStream_t s, int* with_rhs)
{     // *with_rhs (optional) <- 1 when the launch also computed the reduced right-hand side

    if (d.lean && d.v_kf) {
        if (d.nL > 0) hipLaunchKernelGGL(ba_c_kernel, dim3(nblk(d.nL)), dim3(256), 0, s, d, lambda, bad, epoch);
        if (d.n_list > 0 && d.nP > 0) hipLaunchKernelGGL(ba_v_kf_kernel, dim3((d.nP + 3) / 4), dim3(256), 0, s, d, d.n_list);
        if (d.nP <= 0) return;
    }
    else if (d.lean) { if (d.nfree_edges > 0) hipLaunchKernelGGL(ba_v_lean_kernel, dim3(nblk(d.nfree_edges)), dim3(256), 0, s, d, lambda, bad, epoch); if (d.nP <= 0) return; }
    else if (d.nE > 0 && d.nL > 0) hipLaunchKernelGGL(ba_v_kernel, dim3(nblk(d.nE * 6)), dim3(256), 0, s, d, lambda, bad, epoch);
    if (d.row_schur) {
        static bool attr_set[64] = {};
        if (d.row_stream) hipLaunchKernelGGL(ba_schur_row_stream_kernel, dim3(8 * ((d.n_wg + 7) / 8)), dim3(64 * BA_ROW_WAVES), (size_t)(BA_ROW_RANGE + 1) * 144, s, d);
        else {
        ba_opt_in_lds(ba_schur_row_kernel, (int)BA_ROW_LDS, attr_set);
        hipLaunchKernelGGL(ba_schur_row_kernel, dim3(8 * ((d.n_wg + 7) / 8)), dim3(64 * BA_ROW_WAVES), BA_ROW_LDS, s, d);
        }
        const int nblk_blocks = (int)(((size_t)d.nu * 36 + 255) / 256);
        hipLaunchKernelGGL(ba_schur_combine_kernel, dim3(nblk_blocks + (6 * d.nP + 255) / 256), dim3(256), 0, s, d, lambda, nblk_blocks);
        if (with_rhs) *with_rhs = 1;
        return;
    }
    if (d.nu <= BA_SMALL_SPLIT_MAX_UNITS) {
        if (with_rhs && d.lean && d.nu > 0 && d.nP > 0 && d.nP <= 128) { hipLaunchKernelGGL(ba_schur_rhs_split_kernel, dim3(d.nu + d.nP), dim3(64 * BA_SMALL_SPLIT), 0, s, d, lambda); *with_rhs = 1; }
        else if (d.nu > 0) hipLaunchKernelGGL(ba_schur_mfma_kernel<BA_SMALL_SPLIT>, dim3(d.nu), dim3(64 * BA_SMALL_SPLIT), 0, s, d, lambda);
    }
    else hipLaunchKernelGGL(ba_schur_mfma_kernel<1>, dim3(8 * (((d.nu + BA_SCHUR_WAVES - 1) / BA_SCHUR_WAVES + 7) / 8)), dim3(64 * BA_SCHUR_WAVES), 0, s, d, lambda);
}
void ba_launch_row_structure(const CorbBADev& d, hipStream_t s)            // before the pair lists: the rows' block ranges
{
    hipLaunchKernelGGL(ba_urow_kernel, dim3((d.nu + 1 + 255) / 256), dim3(256), 0, s, d);
}
void ba_launch_rr_count(const CorbBADev& d, hipStream_t s)                 // ranges per keyframe and their table sizes, scanned (rr_off[nP], rowwb[nP] = the totals)
{
    hipLaunchKernelGGL(ba_rr_count_kernel, dim3((d.nP + 1 + 255) / 256), dim3(256), 0, s, d);
    hipLaunchKernelGGL(ba_scan_inplace_kernel, dim3(1), dim3(1024), 0, s, d.rr_off, d.nP);
    hipLaunchKernelGGL(ba_scan_inplace_kernel, dim3(1), dim3(1024), 0, s, d.rowwb, d.nP);
}
void ba_launch_rr_stream(const CorbBADev& d, bool fill, hipStream_t s)     // after the units: units -> wavefronts + rounds per wavefront (wave_off, to be scanned by the caller), then the padded streams
{
    if (!fill) hipLaunchKernelGGL(ba_rr_assign_kernel, dim3((d.n_wg + 1 + 255) / 256), dim3(256), 0, s, d);
    else hipLaunchKernelGGL(ba_rr_stream_kernel, dim3((d.n_wg * BA_ROW_WAVES + 3) / 4), dim3(256), 0, s, d);
}
void ba_launch_rr_units(const CorbBADev& d, bool fill, hipStream_t s)      // after the pair lists: unit counts per (workgroup, block) scanned into first-unit indices (wb_unit[n_wb] = the total), then units + headers
{
    if (!fill) {
        hipLaunchKernelGGL(ba_rr_units_kernel<false>, dim3((d.nu + 255) / 256), dim3(256), 0, s, d);
        if (d.n_wb > BA_SCAN_ONE_WG && d.scan_scratch) corb_launch_exclusive_scan(d.wb_unit, d.wb_unit, (size_t)d.n_wb, d.scan_scratch, s);
        else hipLaunchKernelGGL(ba_scan_inplace_kernel, dim3(1), dim3(1024), 0, s, d.wb_unit, d.n_wb);
    } else {
        hipLaunchKernelGGL(ba_rr_units_kernel<true>, dim3((d.nu + 255) / 256), dim3(256), 0, s, d);
        hipLaunchKernelGGL(ba_rr_header_kernel, dim3((d.nP + 255) / 256), dim3(256), 0, s, d);
    }
}

// Block-Jacobi blocks of G = 8 / 16 poses (48 x 48 / 96 x 96): the dense diagonal block of S is inverted IN REGISTERS by the symmetric sweep operator.
// G x G threads, thread (ty, tx) holds the 6 x 6 block (pose k0 + ty, pose k0 + tx) -- exactly one BSR block, found by a binary search in the row's
// columns -- and step k = 0 .. n-1 is   c = column k,  p = 1 / c_k,  a_ij -= c_i c_j p,  a_ik = a_ki = c_i p,  a_kk = -p :  after n steps a = -A^-1 (the pivots
// are those of the Cholesky factorisation: a pivot <= 0 reports "not positive definite" like a failed potrf).  c travels through 768 bytes of LDS (double-buffered:
// one workgroup barrier per step); only values held by threads ON OR ABOVE the diagonal ever feed c, and the result is written from those and mirrored, so the
// inverse is exactly symmetric.  Round 3's form (factorisation by one wavefront in 74.5 KB of LDS, triangular inverse, X' X: a 380 us dependent chain per block
// and two blocks per CU) took 3.5 ms for the 3 125 + 450 blocks of a 50 000-keyframe map; this one is bound by its 42 FP64 instructions per step and thread.
template <int G>
__device__ __forceinline__ void ba_pc_sweep_body(const int nP, const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                 float* __restrict__ out32, int* fail_flag, const int b, double* cbuf, float* stage)
{
    constexpr int n = 6 * G, SP = n + 1;                     // stage pitch (floats)
    const int tid = threadIdx.x, ty = tid / G, tx = tid - ty * G;
    const int ki = b * G + ty, kj = b * G + tx;
    double a[6][6];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int s = 0; s < 6; s++) a[r][s] = 0.0;
    if (ki < nP && kj < nP) {
        int lo = rowptr[ki]; const int end = rowptr[ki + 1]; int hi = end;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (col[mid] < kj) lo = mid + 1; else hi = mid; }
        if (lo < end && col[lo] == kj) {
            const double2* v = reinterpret_cast<const double2*>(val + (size_t)lo * 36);
#pragma unroll
            for (int e = 0; e < 18; e++) { const double2 q = v[e]; a[(2 * e) / 6][(2 * e) % 6] = q.x; a[(2 * e + 1) / 6][(2 * e + 1) % 6] = q.y; }
        }
    } else if (ki == kj) {
#pragma unroll
        for (int r = 0; r < 6; r++) a[r][r] = 1.0;          // padding rows of the last block
    }
    bool fail = false;
    for (int kb = 0; kb < G; kb++) {
#pragma unroll
        for (int kk = 0; kk < 6; kk++) {
            double* cb = cbuf + (kk & 1) * n;                // step k = 6 kb + kk, buffer k & 1
            // column k, from the upper triangle: rows above the diagonal block from column kk of the threads (ty < kb, kb), the rest from row kk of (kb, tx >= kb)
            if (tx == kb && ty < kb) {
#pragma unroll
                for (int r = 0; r < 6; r++) cb[6 * ty + r] = a[r][kk];
            }
            if (ty == kb && tx > kb) {
#pragma unroll
                for (int s = 0; s < 6; s++) cb[6 * tx + s] = a[kk][s];
            }
            if (ty == kb && tx == kb) {
#pragma unroll
                for (int r = 0; r < 6; r++) cb[6 * kb + r] = r <= kk ? a[r][kk] : a[kk][r];
            }
            __syncthreads();
            const double dk = cb[6 * kb + kk];
            double p = 1.0;
            if (dk > 0) p = 1.0 / dk; else fail = true;
            double ci[6], cj[6];
#pragma unroll
            for (int r = 0; r < 6; r++) { ci[r] = cb[6 * ty + r]; cj[r] = cb[6 * tx + r] * p; }
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int s = 0; s < 6; s++) a[r][s] = fma(-ci[r], cj[s], a[r][s]);
            if (ty == kb) {
#pragma unroll
                for (int s = 0; s < 6; s++) a[kk][s] = cj[s];
            }
            if (tx == kb) {
#pragma unroll
                for (int r = 0; r < 6; r++) a[r][kk] = ci[r] * p;
            }
            if (ty == kb && tx == kb) a[kk][kk] = -p;
        }
    }
    if (fail && tid == 0) *fail_flag = 1;                    // not positive definite: the solve fails like a failed potrf
    // A^-1 = -a: the blocks on / above the diagonal into the LDS stage, mirrored; then whole rows out (16-byte stores)
    if (ty <= tx) {
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int s = 0; s < 6; s++) {
                if (ty == tx && s < r) continue;
                const float v = (float)(-a[r][s]);
                stage[(6 * ty + r) * SP + 6 * tx + s] = v; stage[(6 * tx + s) * SP + 6 * ty + r] = v;
            }
    }
    __syncthreads();
    float* o = out32 + (size_t)b * n * n;
    for (int t = tid; t < n * n / 4; t += G * G) {
        const int row = (4 * t) / n, c0 = 4 * t - row * n;
        const float* sr = stage + row * SP + c0;
        *reinterpret_cast<float4*>(o + 4 * (size_t)t) = make_float4(sr[0], sr[1], sr[2], sr[3]);
    }
}
#define BA_PC_SWEEP_LDS(G) (sizeof(double) * 2 * 6 * (G) + sizeof(float) * 6 * (G) * (6 * (G) + 1))
template <int G>
__global__ __launch_bounds__(G * G) void ba_pc_invert_kernel(CorbBADev d)
{
    extern __shared__ double pci_sm[];                      // c[2][n] | stage[n][n + 1] (floats)
    ba_pc_sweep_body<G>(d.nP, d.bsr_rowptr, d.bsr_col, d.bsr_val, d.pc_inv32, d.cg_flag + 1, blockIdx.x, pci_sm, reinterpret_cast<float*>(pci_sm + 2 * 6 * G));
}
// the blocks of the fine level and of every coarse level of the multilevel preconditioner in ONE launch
__global__ __launch_bounds__(BA_ML_G * BA_ML_G) void ba_pc_invert_all_kernel(CorbBADev d, BAMLDev m)
{
    extern __shared__ double pci_sm[];
    float* stage = reinterpret_cast<float*>(pci_sm + 2 * 6 * BA_ML_G);
    if ((int)blockIdx.x < d.pc_nblk) { ba_pc_sweep_body<BA_ML_G>(d.nP, d.bsr_rowptr, d.bsr_col, d.bsr_val, d.pc_inv32, d.cg_flag + 1, blockIdx.x, pci_sm, stage); return; }
    const int bb = blockIdx.x - d.pc_nblk;
    int k = 0;
    while (k + 1 < m.L && bb >= m.lv[k + 1].blk_off) k++;
    const BAMLLevel& c = m.lv[k];                           // the level as a reduced system of its own
    ba_pc_sweep_body<BA_ML_G>(c.n, c.rowptr, c.col, c.val, c.pc_inv32, d.cg_flag + 1, bb - c.blk_off, pci_sm, stage);
}

// pc_refresh = 0: keep the preconditioner blocks of an earlier trial (any symmetric positive definite M is a valid preconditioner)
// the block preconditioner follows S: the dense diagonal blocks (8 or 16 keyframes: 48 / 96 rows) are gathered and inverted in registers by one workgroup each
// (round 2 still sent 32- and 64-keyframe blocks through rocSOLVER's batched potrf / potri; they bought 4 % at 1 200 keyframes)
int ba_launch_pc_refresh(const CorbBADev& d, hipStream_t s)
{
    if (d.nP <= 0 || d.pc_g <= 1) return 0;
    if (d.pc_g != 8 && d.pc_g != 16) return 1;
    if (d.ml && d.pc_g == BA_ML_G) {                    // the coarse levels follow S like the fine blocks do: Galerkin matrices, then all blocks at once
        ba_ml_launch_setup(d, *d.ml, s);
        hipLaunchKernelGGL(ba_pc_invert_all_kernel, dim3(d.pc_nblk + d.ml->n_blocks), dim3(BA_ML_G * BA_ML_G), BA_PC_SWEEP_LDS(BA_ML_G), s, d, *d.ml);
    }
    else if (d.pc_g == 16) hipLaunchKernelGGL(ba_pc_invert_kernel<16>, dim3(d.pc_nblk), dim3(256), BA_PC_SWEEP_LDS(16), s, d);
    else hipLaunchKernelGGL(ba_pc_invert_kernel<8>, dim3(d.pc_nblk), dim3(64), BA_PC_SWEEP_LDS(8), s, d);
    if (d.pc_pack32) hipLaunchKernelGGL(ba_pc_pack_kernel, dim3(d.pc_nblk), dim3(256), 0, s, d);
    return 0;
}
int ba_launch_schur_bsr(const CorbBADev& d, double lambda, int nnzb, int* bad, int epoch, hipStream_t s, int pc_refresh)
{
    (void)hipMemsetAsync(d.cg_flag, 0, 2 * sizeof(int), s);
    if (d.cg_two_level) (void)hipMemsetAsync(d.cg_tick, 0, sizeof(int) * (size_t)(d.cg_ngrp + d.cg_ngrp_spmv + 2) * CG_TICK_STRIDE, s);
    // deterministic MFMA form: no memset, no diagonal / mirror pass -- every block is stored once
    if (d.nL > 0 && !d.lean) hipLaunchKernelGGL(ba_schur_prepare_kernel, dim3(nblk(d.nL)), dim3(256), 0, s, d, lambda, bad, epoch);
    int with_rhs = 0;
    if (d.nP > 0 || d.lean) ba_schur_mfma_launch(d, lambda, bad, epoch, s, d.row_schur ? &with_rhs : nullptr);
    if (d.nP > 0) {
        if (!with_rhs) ba_launch_reduced_rhs(d, s);
        if (d.pc_g <= 1) hipLaunchKernelGGL(ba_minv_kernel, dim3(nblk(d.nP)), dim3(256), 0, s, d);
        else if (pc_refresh) return ba_launch_pc_refresh(d, s);
    }
    return 0;
}
// the solve in progress goes on to a tighter tolerance: the state a stopped solve holds is the state its next iteration starts from (see corb_ba.cpp cg_run)
__global__ void ba_pcg_resume_kernel(CorbBADev d, double tol2) { CG_TOL2(d) = tol2; d.cg_flag[0] = 0; }
void ba_launch_pcg_resume(const CorbBADev& d, double tol, hipStream_t s) { hipLaunchKernelGGL(ba_pcg_resume_kernel, dim3(1), dim3(1), 0, s, d, tol * tol); }
void ba_launch_pcg_init(const CorbBADev& d, double tol, hipStream_t s)
{
    if (d.pc_g > 1) hipLaunchKernelGGL(ba_pcg_init_big_kernel, dim3(d.cg_nparts), dim3(256), sizeof(double) * d.pc_gb, s, d);
    else hipLaunchKernelGGL(ba_pcg_init_kernel, dim3(d.cg_nparts), dim3(256), 0, s, d);
    hipLaunchKernelGGL(ba_pcg_zero_x_kernel, dim3(std::max(1, std::min(256, (d.sp + 4095) / 4096))), dim3(256), 0, s, d, tol * tol);
    if (d.ml && d.pc_g > 1) ba_ml_launch_apply(d, *d.ml, 0, 0, 1, s);      // z0 = M^-1 r0 with the coarse levels; r.z into both parity slots like the init kernel's
}
// `n_iter` (even) CG iterations starting at even parity + the convergence check; graph-capturable.  With the multilevel preconditioner an iteration is four dependent
// launches: SpMV, [step kernel + restriction] (ba_pcg_step_restrict_kernel), coarse block solves, prolongation.  (Round 4 first ran the restriction + coarse solves on a
// second stream beside the step kernel, the captured graph carrying both branches: 181 -> 172 us per iteration at 50 000 keyframes, but a graph's cross-stream edges
// cost ~20 us per iteration -- 1 200 keyframes: 20.7 ms of solve per 10 LM iterations against 11.8 on one stream, crossover near 30 000 -- and the one-launch form
// matches it at 50 000 within 0.4 % (203.6 vs 202.7 ms per 10 LM iterations) and is faster everywhere below: 4 800 keyframes 39.1 -> 35.3 ms, 1 200: 14.6 -> 12.9.)
// par0: parity of the first iteration (a continuation after an odd number of iterations: corb_ba.cpp cg_run)
void ba_launch_pcg_chunk(const CorbBADev& d, int n_iter, hipStream_t s, int par0)
{
    const bool ml = d.ml && d.pc_g > 1;
    for (int t = 0; t < n_iter; t++) {
        const int par = (par0 + t) & 1;
        hipLaunchKernelGGL(ba_pcg_spmv_kernel, dim3(d.cg_nparts_spmv), dim3(256), 0, s, d, par);
        if (ml) ba_ml_launch_step_coarse(d, *d.ml, par, s);
        else if (d.pc_g > 1) hipLaunchKernelGGL(ba_pcg_step_big_kernel, dim3(d.cg_nparts), dim3(256), sizeof(double) * 4 * d.pc_gb, s, d, par);
        else hipLaunchKernelGGL(ba_pcg_step_kernel, dim3(d.cg_nparts), dim3(256), 0, s, d, par);
        if (ml) ba_ml_launch_prolong(d, *d.ml, par, s);      // the new residual is r[par ^ 1]; r.z of iteration parity par
    }
    hipLaunchKernelGGL(ba_pcg_check_kernel, dim3(1), dim3(256), 0, s, d, (par0 + n_iter - 1) & 1);
}

// ------------------------------------------------------------------------------------------------
// Multilevel preconditioner (ba_multilevel.h): Galerkin matrices, block inverses, restriction / block solves / prolongation.
// ------------------------------------------------------------------------------------------------
// A_c = P' A_f P as a GATHER: one wavefront per coarse block (I, J).  Its sources are the fine blocks (i, j) with i under the hat of I and j under the hat of J, and
// because a hat covers a contiguous range of at most 16 fine nodes [lo, hi] they are ONE run of at most 16 slots in each of at most 16 sorted fine rows:
//   1. lanes = fine rows find the runs by binary search side by side (five dependent trips);
//   2. lanes = the 256 (row, entry) pairs, four per lane, fetch the columns and their hat weights (two trips) and compact the pairs with a weight, in
//      (row, entry) order, into a (slot, w_i w_j) list in LDS (ballots + lane-prefix counts);
//   3. lanes = 3 list entries x 18 double2 pieces of a 6 x 6 block add  w a  with eight loads in flight; the three entry slots meet in a fixed order.
// Deterministic.  (Round 4's first form -- a workgroup per coarse ROW scattering every fine entry into LDS copies of the row's blocks, eight 288-byte loads in
// flight per wavefront -- took 2.9 ms for the first level of a 50 000-keyframe map, 280 GB/s; a gather that walked row by row, three trips each, 1.44 ms.)
#ifndef ML_GAL_WAVES
#define ML_GAL_WAVES 8         // (2 / 4 / 8 / 16 wavefronts per coarse row: 274 / 202 / 177 / 177 us per launch on average at 50 000 keyframes)
#endif
__device__ __forceinline__ int gal_mbcnt(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__global__ __launch_bounds__(64 * ML_GAL_WAVES) void ml_galerkin_kernel(const int* __restrict__ f_rowptr, const int* __restrict__ f_col, const double* __restrict__ f_val, BAMLLevel c)
{
    __shared__ int l_slot[ML_GAL_WAVES][256];
    __shared__ double l_w[ML_GAL_WAVES][256];
    const int I = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = c.rowptr[I], c1 = c.rowptr[I + 1];
    const int ilo = c.lo[I], nrow = c.hi[I] - ilo + 1;       // fine rows under the hat of I (<= 16: checked when the hierarchy is built)
    const int rl = lane & 15, rq = lane >> 4;                 // every lane keeps row rl's data; pair (row 4 g + rq, entry rl) in round g
    double wi = 0.0; int r0 = 0, r1 = 0;
    if (rl < nrow) {
        const int i = ilo + rl; const double w1i = c.w1[i];
        wi = (I == c.i0[i] ? 1.0 - w1i : 0.0) + (I == c.i1[i] ? w1i : 0.0);
        r0 = f_rowptr[i]; r1 = f_rowptr[i + 1];
    }
    const int q = lane / 18, el = lane - 18 * q;              // value phase: entry slot q (3 = idle lanes), double2 piece el
    int* my_slot = l_slot[wave]; double* my_w = l_w[wave];
#define GAL_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
    for (int cs = c0 + wave; cs < c1; cs += ML_GAL_WAVES) {
        const int J = c.col[cs], jl = c.lo[J], jh = c.hi[J];
        int s0;                                               // first slot of row rl with column >= jl
        { int a = r0, b = r1; while (a < b) { const int mid = (a + b) >> 1; if (f_col[mid] < jl) a = mid + 1; else b = mid; } s0 = a; }
        int sl[4]; double w[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int r = 4 * g + rq;
            const int sr = __shfl(s0, r), er = __shfl(r1, r); const double wr = __shfl(wi, r);
            sl[g] = sr + rl; w[g] = 0.0;
            if (wr != 0.0 && sl[g] < er) {
                const int j = f_col[sl[g]];
                if (j <= jh) { const double w1j = c.w1[j]; w[g] = wr * ((J == c.i0[j] ? 1.0 - w1j : 0.0) + (J == c.i1[j] ? w1j : 0.0)); }
            }
        }
        GAL_WAVE_SYNC();                                      // (the previous block's list has been read)
        int n = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const unsigned long long bm = __ballot(w[g] != 0.0);
            if (w[g] != 0.0) { const int pos = n + gal_mbcnt(bm); my_slot[pos] = sl[g]; my_w[pos] = w[g]; }
            n += __popcll(bm);
        }
        GAL_WAVE_SYNC();
        double2 acc = make_double2(0.0, 0.0);
        for (int t0 = 0; 3 * t0 < n; t0 += 8) {
            double2 v[8]; double ww[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int pp = 3 * (t0 + k) + q; const bool ok = q < 3 && pp < n;
                const int s = my_slot[ok ? pp : 0]; ww[k] = ok ? my_w[pp] : 0.0;
                v[k] = reinterpret_cast<const double2*>(f_val + (size_t)s * 36)[el < 18 ? el : 0];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) { acc.x += ww[k] * v[k].x; acc.y += ww[k] * v[k].y; }
        }
        const double x1 = __shfl(acc.x, lane + 18), y1 = __shfl(acc.y, lane + 18), x2 = __shfl(acc.x, lane + 36), y2 = __shfl(acc.y, lane + 36);
        if (lane < 18) reinterpret_cast<double2*>(c.val + (size_t)cs * 36)[lane] = make_double2((acc.x + x1) + x2, (acc.y + y1) + y2);
    }
#undef GAL_WAVE_SYNC
}
// chunk sums of r_k = W_k r for the nodes of all levels: one wavefront per chunk of a node's (keyframe, weight) list, fixed-order lane sum.
// q != nullptr (inside a CG iteration): r is the iteration's NEW residual r[par] - alpha q, formed here from the same scalars, with the same two operations and
// therefore the same bits as ba_pcg_step_big_kernel forms it -- the restriction does not wait for that kernel: its workgroups ride in the same launch
// (ba_pcg_step_restrict_kernel).
__device__ __forceinline__ void ml_restrict_body(const CorbBADev& d, const BAMLDev& m, const double* r, const double* q, int par, const int vbid)
{
    // one batch of loads: the flags, the iteration's scalars, the chunk's range (a test between them would cost a memory round trip each)
    const int c = vbid * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cc = c < m.n_chunks ? c : 0;
    const int f1 = d.cg_flag[1], f0 = d.cg_flag[0];
    const int e_begin = m.ch_begin[cc], e_end = m.ch_begin[cc + 1];
    double alpha = 0.0, rr_prev = 0, rz = 0, pq = 0, tol2 = 0, bb = 0;
    if (q) { rr_prev = *CG_FIN_RR(d, par ^ 1); rz = *CG_FIN_RZ(d, par ^ 1); pq = *CG_FIN_PQ(d); tol2 = CG_TOL2(d); bb = d.cg_scal[2]; }
    if (f1 || f0) return;
    if (q) {                                                  // the step kernel's own early-outs, then its alpha
        if (rr_prev <= tol2 * bb || !(pq > 0)) return;
        alpha = rz / pq;
    }
    if (c >= m.n_chunks) return;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int e = e_begin + lane; e < e_end; e += 64) {
        const double w = m.r_w[e]; const double* rp = r + 6 * (size_t)m.r_pose[e];
        if (q) {
            const double* qp = q + 6 * (size_t)m.r_pose[e];
#pragma unroll
            for (int a = 0; a < 6; a++) acc[a] += w * (rp[a] - alpha * qp[a]);
        } else {
#pragma unroll
            for (int a = 0; a < 6; a++) acc[a] += w * rp[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        double v = acc[a];
        v = lx_wave_sum(v);
        if (lane == 0) m.ch_sum[6 * (size_t)c + a] = v;
    }
}
__global__ __launch_bounds__(256) void ml_restrict_kernel(CorbBADev d, BAMLDev m, const double* r, const double* q, int par)
{
    ml_restrict_body(d, m, r, q, par, (int)blockIdx.x);
}
// ONE launch for the step kernel's workgroups and, behind them, the restriction's (it needs nothing the step kernel writes): a CG iteration is then four
// dependent launches instead of five -- what the iteration costs below ~30 000 keyframes, where every launch is 5-13 us of mostly fixed overhead.
__global__ __launch_bounds__(256) void ba_pcg_step_restrict_kernel(CorbBADev d, BAMLDev m, int par)
{
    __shared__ double red[20];
    __shared__ int cnt;
    __shared__ double pc_yw[4 * 96];
    extern __shared__ double pc_rn[];                      // [4][pc_gb]
    if ((int)blockIdx.x >= d.cg_nparts) { ml_restrict_body(d, m, d.cg_r[par], d.cg_q, par, (int)blockIdx.x - d.cg_nparts); return; }
    if (d.pc_pack32) { if (d.pc_gb == 96) ba_pcg_step_sym_body<6>(d, par, pc_rn, pc_yw, red, &cnt); else ba_pcg_step_sym_body<3>(d, par, pc_rn, pc_yw, red, &cnt); }
    else
    if (d.pc_inv32) ba_pcg_step_big_body<float>(d, d.pc_inv32, par, pc_rn, red, &cnt);
    else ba_pcg_step_big_body<double>(d, d.pc_inv, par, pc_rn, red, &cnt);
}
// y_k = D_k^-1 r_k: one workgroup per block-Jacobi block of any level (the inverse is symmetric: thread t reads column t, consecutive addresses).  Four threads per
// row, a quarter of the columns each (a 24-term chain instead of 96: the launch is a few hundred workgroups, i.e. latency), the quarters added in order.
#define ML_APPLY_Q 4
__global__ __launch_bounds__(6 * BA_ML_G * ML_APPLY_Q) void ml_apply_kernel(CorbBADev d, BAMLDev m)
{
    __shared__ double rn[6 * BA_ML_G];
    __shared__ double part[ML_APPLY_Q][6 * BA_ML_G];
    int k = 0;
    while (k + 1 < m.L && (int)blockIdx.x >= m.lv[k + 1].blk_off) k++;
    const BAMLLevel& lv = m.lv[k];
    constexpr int n = 6 * BA_ML_G, nq = n / ML_APPLY_Q;
    const int b = blockIdx.x - lv.blk_off, tid = threadIdx.x, q = tid / n, t = tid - q * n;
    const int row0 = 6 * (lv.node_off + b * BA_ML_G), rows = min(n, 6 * (lv.n - b * BA_ML_G));
    // one batch of loads: the inverse block's operands (they do not depend on r_k), the node's chunk range, the flags -- a test of the flags first costs a round trip
    const float* D = lv.pc_inv32 + (size_t)b * n * n + (size_t)q * nq * n + t;
    float dv[nq];
#pragma unroll
    for (int c = 0; c < nq; c++) dv[c] = D[(size_t)c * n];
    const int g_ = (q == 0 && t < rows) ? (row0 + t) / 6 : 0;
    const int c_first = m.ch_ptr[g_], c_last = m.ch_ptr[g_ + 1];
    const int f1 = d.cg_flag[1], f0 = d.cg_flag[0];
    if (f1 || f0) return;
    if (q == 0) {
        double v = 0;
        if (t < rows) { const int a = (row0 + t) - 6 * g_; for (int c = c_first; c < c_last; c++) v += m.ch_sum[6 * (size_t)c + a]; }      // the node's chunks, in order
        rn[t] = v;
    }
    __syncthreads();
    double acc = 0;
#pragma unroll
    for (int c = 0; c < nq; c++) acc += (double)dv[c] * rn[q * nq + c];
    part[q][t] = acc;
    __syncthreads();
    if (q == 0 && t < rows) m.yk[row0 + t] = lv.wgt * (((part[0][t] + part[1][t]) + part[2][t]) + part[3][t]);
}
// z += sum_k W_k' y_k; r.z of the full preconditioner (workgroup partials, then the three-level tree of the CG kernels) into the final slot(s)
__global__ __launch_bounds__(256) void ml_prolong_kernel(CorbBADev d, BAMLDev m, const double* r, int par, int both)
{
    __shared__ double red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int pose_ = i < d.sp ? i / 6 : 0;
    const int f1 = d.cg_flag[1], f0 = d.cg_flag[0];        // (one batch with the list's range: a test of the flags first costs a memory round trip)
    const int e_first = m.p_ptr[pose_], e_last = m.p_ptr[pose_ + 1];
    if (f1 || f0) return;
    double rz = 0;
    if (i < d.sp) {
        const int pose = i / 6, a = i - 6 * pose;
        // (four entries in flight: the loop is a chain of gathers otherwise -- 14 us for ~30 entries per keyframe at 50 000 keyframes)
        double z0 = 0, z1 = 0, z2 = 0, z3 = 0;
        int e = e_first; const int e1 = e_last;
        for (; e + 4 <= e1; e += 4) {
            const int n0 = m.p_node[e], n1 = m.p_node[e + 1], n2 = m.p_node[e + 2], n3 = m.p_node[e + 3];
            const double w0 = m.p_w[e], w1 = m.p_w[e + 1], w2 = m.p_w[e + 2], w3 = m.p_w[e + 3];
            z0 += w0 * m.yk[6 * (size_t)n0 + a]; z1 += w1 * m.yk[6 * (size_t)n1 + a]; z2 += w2 * m.yk[6 * (size_t)n2 + a]; z3 += w3 * m.yk[6 * (size_t)n3 + a];
        }
        for (; e < e1; e++) z0 += m.p_w[e] * m.yk[6 * (size_t)m.p_node[e] + a];
        const double zc = (z0 + z1) + (z2 + z3);
        const double z = d.cg_z[i] + zc;
        d.cg_z[i] = z;
        rz = r[i] * z;
    }
    const double s1 = block_sum_256(rz, red);
    if (threadIdx.x == 0) cg_publish(&m.part[blockIdx.x], s1);
    if (threadIdx.x < 64)
        cg_tree_reduce(m.tick, m.tick + (size_t)m.ngrp * CG_TICK_STRIDE, m.ngrp, m.part, nullptr, m.part2, nullptr, nullptr, nullptr,
                       CG_FIN_RZ(d, par), both ? CG_FIN_RZ(d, par ^ 1) : nullptr, nullptr, nullptr);
}
void ba_ml_launch_setup(const CorbBADev& d, const BAMLDev& m, hipStream_t s)
{
    (void)hipMemsetAsync(m.tick, 0, sizeof(int) * (size_t)(m.ngrp + 1) * CG_TICK_STRIDE, s);
    for (int k = 0; k < m.L; k++) {                              // A_k = P' A_{k-1} P, level by level; the blocks of all levels are inverted by the caller's one launch
        const BAMLLevel& c = m.lv[k];
        const int* f_rowptr = k == 0 ? d.bsr_rowptr : m.lv[k - 1].rowptr; const int* f_col = k == 0 ? d.bsr_col : m.lv[k - 1].col;
        const double* f_val = k == 0 ? d.bsr_val : m.lv[k - 1].val;
        hipLaunchKernelGGL(ml_galerkin_kernel, dim3(c.n), dim3(64 * ML_GAL_WAVES), 0, s, f_rowptr, f_col, f_val, c);
    }
}
void ba_ml_launch_apply(const CorbBADev& d, const BAMLDev& m, int r_buf, int par, int both, hipStream_t s)
{
    const double* r = d.cg_r[r_buf];
    hipLaunchKernelGGL(ml_restrict_kernel, dim3((m.n_chunks + 3) / 4), dim3(256), 0, s, d, m, r, (const double*)nullptr, 0);
    hipLaunchKernelGGL(ml_apply_kernel, dim3(m.n_blocks), dim3(6 * BA_ML_G * ML_APPLY_Q), 0, s, d, m);
    hipLaunchKernelGGL(ml_prolong_kernel, dim3(m.np), dim3(256), 0, s, d, m, r, par, both);
}
// inside CG iteration `par`: the step kernel's launch carries the restriction (ba_pcg_step_restrict_kernel), then the coarse block solves
void ba_ml_launch_step_coarse(const CorbBADev& d, const BAMLDev& m, int par, hipStream_t s)
{
    hipLaunchKernelGGL(ba_pcg_step_restrict_kernel, dim3(d.cg_nparts + (m.n_chunks + 3) / 4), dim3(256), sizeof(double) * 4 * d.pc_gb, s, d, m, par);
    hipLaunchKernelGGL(ml_apply_kernel, dim3(m.n_blocks), dim3(6 * BA_ML_G * ML_APPLY_Q), 0, s, d, m);
}
void ba_ml_launch_prolong(const CorbBADev& d, const BAMLDev& m, int par, hipStream_t s)
{
    hipLaunchKernelGGL(ml_prolong_kernel, dim3(m.np), dim3(256), 0, s, d, m, (const double*)d.cg_r[par ^ 1], par, 0);
}

// e->computeError(); e->chi2(); isDepthPositive() for every edge (types_six_dof_expmap.h:90-103, 122-135)
__global__ __launch_bounds__(256) void ba_edge_eval_kernel(CorbBADev d, double* chi2, double* depth)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.nE) return;
    double err[3], Xc[3];
    chi2[i] = edge_error(d, i, err, Xc);
    depth[i] = Xc[2];
}
void ba_launch_edge_eval(const CorbBADev& d, double* chi2, double* depth, hipStream_t s)
{
    hipLaunchKernelGGL(ba_edge_eval_kernel, dim3(nblk(d.nE)), dim3(256), 0, s, d, chi2, depth);
}
