// graph_kernels.hip -- Optimizer::OptimizeEssentialGraph (C/src/Optimizer.cc:840-1117) on gfx950: pose graph over Sim3 vertices.
//   eg_chi2_kernel      activeChi2: sum |log(C * Si * Sj^-1)|^2 over the edges (identity information, no robust kernel)
//   eg_build_kernel     one thread per edge: error, numeric Jacobians of BOTH vertices (g2o BaseBinaryEdge::linearizeOplus,
//                       central differences, delta 1e-9, 28 perturbed error evaluations), J'J blocks and -J'e added into the
//                       dense system with fp64 atomics (a keyframe has tens of edges; the system is (7 x free vertices)^2)
//   eg_lambda_kernel    A = H + lambda I (copy for the factorisation), x = b
//   eg_update_kernel    Si <- exp(x_i) * Si  (VertexSim3Expmap::oplusImpl, _fix_scale)
//   eg_apply_kernel     SE3 recovery [R | t/s] and map point correction through the reference keyframe (:1045-1114)
// The factorisation itself is the hand-written blocked Cholesky of dense_chol.hip (launched by corb_graph.cpp); the LM control flow is g2o's with
// setUserLambdaInit(1e-16).  Semantics follow oracle/orc_sim3.c.
#include "graph_internal.h"
#include "lane_exchange.h"
#include "sim3_math.h"


__device__ __forceinline__ void eg_edge_error(const S3State& C, const S3State& Si, const S3State& Sj, double* e)
{
    S3State Sjinv, t1, t2; s3_inv(Sj, Sjinv); s3_mul(C, Si, t1); s3_mul(t1, Sjinv, t2); s3_log(t2, e);
}

__global__ __launch_bounds__(256) void eg_chi2_kernel(CorbGraphDev d)
{
    __shared__ double red[4];
    double acc = 0;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < d.E; e += gridDim.x * 256) {
        const int a = d.vi[e], c = d.vj[e];
        if (d.fixed[a] && d.fixed[c]) continue;
        S3State C, Si, Sj; s3_load(d.meas + 8 * (size_t)e, C); s3_load(d.V + 8 * (size_t)a, Si); s3_load(d.V + 8 * (size_t)c, Sj);
        double er[7]; eg_edge_error(C, Si, Sj, er);
        for (int q = 0; q < 7; q++) acc += er[q] * er[q];
    }
    acc = lx_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) d.partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void eg_reduce_kernel(const double* partial, int n, double* out)
{
    __shared__ double red[4];
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    acc = lx_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(64) void eg_build_kernel(CorbGraphDev d)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= d.E) return;
    const int a = d.vi[e], c = d.vj[e];
    if (d.fixed[a] && d.fixed[c]) return;
    S3State C, Si, Sj; s3_load(d.meas + 8 * (size_t)e, C); s3_load(d.V + 8 * (size_t)a, Si); s3_load(d.V + 8 * (size_t)c, Sj);
    double err[7], Ji[49], Jj[49];
    eg_edge_error(C, Si, Sj, err);
    const double scalar = 1.0 / (2 * 1e-9);
    for (int side = 0; side < 2; side++) {
        if (d.fixed[side ? c : a]) continue;
        double* J = side ? Jj : Ji;
        for (int dd = 0; dd < 7; dd++) {
            double u[7] = { 0, 0, 0, 0, 0, 0, 0 }, ep[7], em[7];
            S3State P = side ? Sj : Si; u[dd] = 1e-9; s3_oplus(P, u, d.fix_scale);
            if (side) eg_edge_error(C, Si, P, ep); else eg_edge_error(C, P, Sj, ep);
            P = side ? Sj : Si; u[dd] = -1e-9; s3_oplus(P, u, d.fix_scale);
            if (side) eg_edge_error(C, Si, P, em); else eg_edge_error(C, P, Sj, em);
            for (int r = 0; r < 7; r++) J[r * 7 + dd] = scalar * (ep[r] - em[r]);
        }
    }
    double* o = d.ejac + (size_t)e * 105;
    const bool fa = d.fixed[a], fc = d.fixed[c];
    for (int t = 0; t < 49; t++) { o[t] = fa ? 0.0 : Ji[t]; o[49 + t] = fc ? 0.0 : Jj[t]; }
    for (int r = 0; r < 7; r++) o[98 + r] = err[r];
}

// H(v,v) = sum J'J and b(v) = -sum J'e over the edges incident to free vertex v, in edge order (the oracle's order); lanes 0..48 own one entry of
// the 7x7 block, lanes 49..55 one entry of b.  No atomics: the result does not depend on the scheduling.
__global__ __launch_bounds__(64) void eg_accum_vertex_kernel(CorbGraphDev d)
{
    const int h = blockIdx.x, lane = threadIdx.x;
    if (lane >= 56) return;
    const int p = lane < 49 ? lane / 7 : lane - 49, q = lane < 49 ? lane - 7 * p : 0;
    double acc = 0;
    for (int t = d.voff[h]; t < d.voff[h + 1]; t++) {
        const int e = d.vedge[t] >> 1, role = d.vedge[t] & 1;
        const double* J = d.ejac + (size_t)e * 105 + 49 * role; const double* er = d.ejac + (size_t)e * 105 + 98;
        double v = 0;
        if (lane < 49) { for (int r = 0; r < 7; r++) v += J[r * 7 + p] * J[r * 7 + q]; }
        else { for (int r = 0; r < 7; r++) v += J[r * 7 + p] * (-er[r]); }
        acc += v;
    }
    if (lane < 49) d.H[(size_t)(7 * h + p) * d.sp + 7 * h + q] = acc; else d.b[7 * h + p] = acc;
}
// H(lo,hi) = sum J_lo' J_hi over the edges joining the pair (edge order), H(hi,lo) = its transpose
__global__ __launch_bounds__(64) void eg_accum_pair_kernel(CorbGraphDev d)
{
    const int g = blockIdx.x, lane = threadIdx.x;
    if (lane >= 49) return;
    const int p = lane / 7, q = lane - 7 * p;
    double acc = 0;
    for (int t = d.poff[g]; t < d.poff[g + 1]; t++) {
        const int e = d.pedge[t] >> 1, flip = d.pedge[t] & 1;
        const double* Jlo = d.ejac + (size_t)e * 105 + 49 * flip; const double* Jhi = d.ejac + (size_t)e * 105 + 49 * (1 - flip);
        double v = 0;
        for (int r = 0; r < 7; r++) v += Jlo[r * 7 + p] * Jhi[r * 7 + q];
        acc += v;
    }
    const int lo = d.plo[g], hi = d.phi[g];
    d.H[(size_t)(7 * lo + p) * d.sp + 7 * hi + q] = acc;
    d.H[(size_t)(7 * hi + q) * d.sp + 7 * lo + p] = acc;
}

__global__ __launch_bounds__(256) void eg_lambda_kernel(CorbGraphDev d, double lambda)
{
    const size_t n = (size_t)d.sp * d.sp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t r = i / d.sp, c = i - r * d.sp;
        d.A[i] = d.H[i] + (r == c ? lambda : 0.0);
        if (c == 0) d.x[r] = d.b[r];
    }
}
__global__ __launch_bounds__(256) void eg_scale_kernel(CorbGraphDev d, double lambda, double* out)
{
    __shared__ double red[4];
    double acc = 0;
    for (int j = threadIdx.x; j < d.sp; j += 256) acc += d.x[j] * (lambda * d.x[j] + d.b[j]);
    acc = lx_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(64) void eg_update_kernel(CorbGraphDev d)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= d.K || d.idx[k] < 0) return;
    S3State S; s3_load(d.V + 8 * (size_t)k, S);
    s3_oplus(S, d.x + 7 * (size_t)d.idx[k], d.fix_scale);
    s3_store(S, d.V + 8 * (size_t)k);
}
__global__ __launch_bounds__(256) void eg_apply_kernel(int K, const double* S_old, const double* S_new, float* Tiw, int M, const int* ref, float* points)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < K) {
        S3State S; s3_load(S_new + 8 * (size_t)i, S);
        double R[9]; quat_to_R(S.q, R);
        const double is = 1. / S.s;
        float* T = Tiw + 16 * (size_t)i;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[r * 4 + c] = (float)R[r * 3 + c]; T[r * 4 + 3] = (float)(S.t[r] * is); }
        T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
    }
    if (i < M && ref[i] >= 0 && ref[i] < K) {
        S3State Srw, Snew, Swr; s3_load(S_old + 8 * (size_t)ref[i], Srw); s3_load(S_new + 8 * (size_t)ref[i], Snew); s3_inv(Snew, Swr);
        const double p[3] = { (double)points[3 * (size_t)i], (double)points[3 * (size_t)i + 1], (double)points[3 * (size_t)i + 2] };
        double a[3], c[3]; s3_map(Srw, p, a); s3_map(Swr, a, c);
        points[3 * (size_t)i] = (float)c[0]; points[3 * (size_t)i + 1] = (float)c[1]; points[3 * (size_t)i + 2] = (float)c[2];
    }
}

void eg_launch_chi2(const CorbGraphDev& d, int nparts, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(eg_chi2_kernel, dim3(nparts), dim3(256), 0, s, d);
    hipLaunchKernelGGL(eg_reduce_kernel, dim3(1), dim3(256), 0, s, d.partial, nparts, out);
}
void eg_launch_build(const CorbGraphDev& d, hipStream_t s)
{
    (void)hipMemsetAsync(d.H, 0, sizeof(double) * (size_t)d.sp * d.sp, s);
    (void)hipMemsetAsync(d.b, 0, sizeof(double) * (size_t)d.sp, s);
    if (d.E > 0) hipLaunchKernelGGL(eg_build_kernel, dim3((d.E + 63) / 64), dim3(64), 0, s, d);
    if (d.nP > 0) hipLaunchKernelGGL(eg_accum_vertex_kernel, dim3(d.nP), dim3(64), 0, s, d);
    if (d.n_pairs > 0) hipLaunchKernelGGL(eg_accum_pair_kernel, dim3(d.n_pairs), dim3(64), 0, s, d);
}
void eg_launch_lambda(const CorbGraphDev& d, double lambda, hipStream_t s)
{
    const size_t n = (size_t)d.sp * d.sp;
    const int blocks = (int)((n + 255) / 256 > 65535 ? 65535 : (n + 255) / 256);
    hipLaunchKernelGGL(eg_lambda_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, d, lambda);
}
void eg_launch_update(const CorbGraphDev& d, double lambda, double* scale_out, hipStream_t s)
{
    hipLaunchKernelGGL(eg_scale_kernel, dim3(1), dim3(256), 0, s, d, lambda, scale_out);
    hipLaunchKernelGGL(eg_update_kernel, dim3((d.K + 63) / 64), dim3(64), 0, s, d);
}
void eg_launch_apply(int K, const double* S_old, const double* S_new, float* Tiw, int M, const int* ref, float* points, hipStream_t s)
{
    const int n = K > M ? K : M;
    if (n > 0) hipLaunchKernelGGL(eg_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, s, K, S_old, S_new, Tiw, M, ref, points);
}
