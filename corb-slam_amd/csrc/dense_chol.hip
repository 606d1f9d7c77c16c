// dense_chol.hip -- dense symmetric positive definite solve  A x = b  in FP64 on gfx950, hand-written (replaces rocSOLVER dpotrf + dpotrs on the bundle-adjustment
// path: g2o's LinearSolverEigen / LinearSolverDense on the reduced camera system, G/solvers/linear_solver_eigen.h:94-124, for maps of up to a few hundred free
// keyframes, and the (7 K)^2 system of Optimizer::OptimizeEssentialGraph).
//
// Right-looking blocked Cholesky, panel width NB = 32, on the row-major lower triangle; the right-hand side travels as row n of the matrix, so that the
// forward substitution  y = L^-1 b  falls out of the factorisation's own triangular solves and rank-NB updates:
//   chol_panel_kernel    one wavefront per 64 rows below (and including) the diagonal block of panel k: every wavefront factorises the 32 x 32 diagonal block
//                        itself (cheaper than a launch that would broadcast it; the factor is filed in a WORKSPACE, the block in A is left as it was: see the
//                        kernel), then solves its rows against it:  L_ik = A_ik L_kk^-T
//   chol_update_kernel   trailing update  A_ij -= L_ik L_jk'  for the 64 x 64 tiles i >= j > k on the FP64 matrix cores (v_mfma_f64_16x16x4_f64: a wavefront owns a
//                        32 x 32 quarter, 4 accumulator tiles, panels staged through LDS), and  b_j -= L_jk y_k  on the diagonal tiles
//   chol_backsub_kernel  x = L^-T y  by ONE workgroup walking the panels backwards (the lower triangle is read once, coalesced along the rows)
// Two launches per panel, fixed summation order, no atomics: bit-identical runs.  *info = 0, or 1 + the first column whose pivot is not positive / finite
// (the factor is then meaningless; the LM loop rejects the trial like g2o does when its solver returns false, optimization_algorithm_levenberg.cpp:126-127).
#include "dense_chol.h"

#define CH_NB 32
#define CH_ROWS 64            // rows per wavefront of the panel kernel / tile edge of the update kernel

typedef double chol_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double ch_readlane(double v, int src)
{ return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src)); }

// element (r, c) of the augmented matrix: rows 0..n-1 are A (row-major, leading dimension ld), row n is b
__device__ __forceinline__ double* ch_at(double* A, double* b, int n, int ld, int r, int c) { return r < n ? A + (size_t)r * ld + c : b + c; }

__global__ __launch_bounds__(64) void chol_panel_kernel(double* A, double* b, int n, int ld, int k, int* info, double* diag_ws)
{
    // (measured alternative: the diagonal block in LDS with rolled loops and broadcast reads -- a few hundred bytes of code instead of ~8 000 unrolled
    // instructions -- was 15 % SLOWER per panel: the chain of dependent LDS round trips costs more than the instruction fetch)
    __shared__ double tile[CH_ROWS][CH_NB + 1];
    const int lane = threadIdx.x;
    const int c0 = k * CH_NB, w = min(CH_NB, n - c0);          // this panel's columns [c0, c0 + w)
    // ---- the diagonal block, by every wavefront: lane r (< 32) holds row r of the lower triangle; columns >= w are an identity tail ----
    double L[CH_NB];
    {
        const int r = lane & 31;
#pragma unroll
        for (int c = 0; c < CH_NB; c++) L[c] = (r < w && c < w && c <= r) ? A[(size_t)(c0 + r) * ld + c0 + c] : (r == c ? 1.0 : 0.0);
    }
    int bad = 0;
    double dinv[CH_NB];                                        // 1 / L[c][c], wave-uniform
#pragma unroll
    for (int c = 0; c < CH_NB; c++) {
        // column c: L[r][c] = (a[r][c] - sum_{m<c} L[r][m] L[c][m]) / L[c][c]; row c is broadcast from lane c
        double s = L[c], s1 = 0, s2 = 0, s3 = 0;              // four partial sums: shorter chains of dependent FP64 multiply-adds
#pragma unroll
        for (int m = 0; m < c; m++) {
            const double pr = L[m] * ch_readlane(L[m], c);
            if ((m & 3) == 0) s -= pr; else if ((m & 3) == 1) s1 -= pr; else if ((m & 3) == 2) s2 -= pr; else s3 -= pr;
        }
        s += (s1 + s2) + s3;
        const double piv = ch_readlane(s, c);
        if (!(piv > 0.0) || !isfinite(piv)) { if (!bad) bad = c0 + c + 1; }
        const double lcc = sqrt(piv > 0.0 ? piv : 1.0);
        dinv[c] = 1.0 / lcc;
        L[c] = (lane & 31) == c ? lcc : ((lane & 31) > c ? s * dinv[c] : 0.0);
    }
    const int first = c0 + (int)blockIdx.x * CH_ROWS;          // chunk 0 starts AT the diagonal block (its first rows are the block itself)
    const int nrows = n + 1;                                   // + the right-hand side row
    // The factor of the diagonal block goes to the caller's workspace, NOT back into A: every wavefront of this launch reads the block's ORIGINAL entries above, and a
    // workgroup that the dispatcher starts late -- other streams' kernels on the compute units -- would read a block that workgroup 0 had already overwritten with its
    // factor (found in round 6: a dense-solver BA beside tracking calls on a second stream returned a different chi2 / lambda sequence in ~1 % of the calls, never alone;
    // tools/conc_probe3.py).  The diagonal blocks of A keep their input values; chol_backsub_kernel takes the factors from the workspace.
    if (blockIdx.x == 0 && lane < CH_NB) {
        double* dst = diag_ws + (size_t)k * CH_NB * CH_NB + (size_t)lane * CH_NB;
#pragma unroll
        for (int c = 0; c < CH_NB; c++) dst[c] = L[c];          // (rows / columns >= w: the identity tail)
    }
    if (blockIdx.x == 0 && lane == 0 && bad) atomicCAS(info, 0, bad);
    // rows [first + (chunk 0 ? w : 0), first + 64) of the augmented matrix: in through LDS (lane = column, 2 rows per step), then lane = row
    const int rbeg = first + (blockIdx.x == 0 ? w : 0);
    for (int rr = lane >> 5; rr < CH_ROWS; rr += 2) {
        const int r = first + rr, c = lane & 31;
        tile[rr][c] = (r >= rbeg && r < nrows && c < w) ? *ch_at(A, b, n, ld, r, c0 + c) : 0.0;
    }
    __syncthreads();
    double x[CH_NB];
#pragma unroll
    for (int c = 0; c < CH_NB; c++) x[c] = tile[lane][c];
    // X L_kk' = A_ik  =>  x[c] = (a[c] - sum_{m<c} x[m] L_kk[c][m]) / L_kk[c][c];  L_kk[c][m] lives in lane c's register m
#pragma unroll
    for (int c = 0; c < CH_NB; c++) {
        double s = x[c], s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int m = 0; m < c; m++) {
            const double pr = x[m] * ch_readlane(L[m], c);
            if ((m & 3) == 0) s -= pr; else if ((m & 3) == 1) s1 -= pr; else if ((m & 3) == 2) s2 -= pr; else s3 -= pr;
        }
        s += (s1 + s2) + s3;
        x[c] = s * dinv[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CH_NB; c++) tile[lane][c] = x[c];
    __syncthreads();
    for (int rr = lane >> 5; rr < CH_ROWS; rr += 2) {
        const int r = first + rr, c = lane & 31;
        if (r >= rbeg && r < nrows && c < w) *ch_at(A, b, n, ld, r, c0 + c) = tile[rr][c];
    }
}

// trailing update with panel k: tiles (ti, tj), ti >= tj, of 64 x 64 over the rows / columns from (k + 1) NB on; blockIdx.x enumerates the lower triangle of tiles
__global__ __launch_bounds__(256) void chol_update_kernel(double* A, double* b, int n, int ld, int k)
{
    __shared__ double Li[CH_ROWS][CH_NB + 1];
    __shared__ double Lj[CH_ROWS][CH_NB + 1];
    __shared__ double yk[CH_NB];
    const int c0 = k * CH_NB, w = min(CH_NB, n - c0), t0 = c0 + CH_NB;      // trailing part starts at row / column t0
    int ti = 0; while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ti++;
    const int tj = (int)blockIdx.x - ti * (ti + 1) / 2;
    const int ri = t0 + ti * CH_ROWS, rj = t0 + tj * CH_ROWS;
    for (int e = threadIdx.x; e < CH_ROWS * CH_NB; e += 256) {
        const int rr = e >> 5, c = e & 31;
        Li[rr][c] = (ri + rr < n && c < w) ? A[(size_t)(ri + rr) * ld + c0 + c] : 0.0;
        Lj[rr][c] = (rj + rr < n && c < w) ? A[(size_t)(rj + rr) * ld + c0 + c] : 0.0;
    }
    if (threadIdx.x < CH_NB) yk[threadIdx.x] = threadIdx.x < w ? b[c0 + threadIdx.x] : 0.0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = (wave >> 1) * 32, qj = (wave & 1) * 32;     // this wavefront's 32 x 32 quarter
    // v_mfma_f64_16x16x4_f64: A[i = lane & 15][kk = lane >> 4], B[kk = lane >> 4][j = lane & 15], D[row = (lane >> 4) + 4 reg][col = lane & 15]
    const int li = lane & 15, kk = lane >> 4;
    chol_d4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
#pragma unroll
    for (int k4 = 0; k4 < CH_NB; k4 += 4) {
        const double a0 = Li[qi + li][k4 + kk], a1 = Li[qi + 16 + li][k4 + kk];
        const double b0 = Lj[qj + li][k4 + kk], b1 = Lj[qj + 16 + li][k4 + kk];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int ai = 0; ai < 2; ai++)
#pragma unroll
        for (int bj = 0; bj < 2; bj++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int r = ri + qi + 16 * ai + kk + 4 * reg, c = rj + qj + 16 * bj + li;
                if (r < n && c < n && c <= r) A[(size_t)r * ld + c] -= acc[ai][bj][reg];
            }
    // the right-hand side row:  b_j -= L_jk y_k  for the rows of this tile column (once: on the diagonal tiles)
    if (ti == tj && threadIdx.x < CH_ROWS && rj + (int)threadIdx.x < n) {
        double s = 0;
#pragma unroll
        for (int c = 0; c < CH_NB; c++) s += Lj[threadIdx.x][c] * yk[c];
        b[rj + threadIdx.x] -= s;
    }
}

// x = L^-T y: panels from the last to the first; within a panel the 32 x 32 triangle is solved by one wavefront, then every thread takes columns of the
// rows of that panel to the left of the diagonal block:  y_j -= sum_i L[i][j] x_i  (rows i of the panel are contiguous in j: coalesced)
__global__ __launch_bounds__(1024) void chol_backsub_kernel(const double* A, double* b, int n, int ld, const double* diag_ws)
{
    __shared__ double xk[CH_NB];
    __shared__ double blk[CH_NB][CH_NB + 1];
    const int np = (n + CH_NB - 1) / CH_NB;
    for (int k = np - 1; k >= 0; k--) {
        const int c0 = k * CH_NB, w = min(CH_NB, n - c0);
        { const int rr = threadIdx.x >> 5, c = threadIdx.x & 31; blk[rr][c] = (rr < w && c < w && c <= rr) ? diag_ws[(size_t)k * CH_NB * CH_NB + rr * CH_NB + c] : (rr == c ? 1.0 : 0.0); }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            // L_kk' x = y, from the last unknown up: x_c = y_c / L[c][c], then y_j -= L[c][j] x_c for j < c (lane j holds y_j)
            double y = lane < w ? b[c0 + lane] : 0.0, x = 0;
            for (int c = CH_NB - 1; c >= 0; c--) {
                const double xc = ch_readlane(y, c) / blk[c][c];
                if (lane == c) x = xc;
                if (lane < c) y -= blk[c][lane] * xc;
            }
            if (lane < w) b[c0 + lane] = x;
            if (lane < CH_NB) xk[lane] = lane < w ? x : 0.0;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < c0; j += 1024) {
            double s = 0;
#pragma unroll 8
            for (int i = 0; i < w; i++) s += A[(size_t)(c0 + i) * ld + j] * xk[i];
            b[j] -= s;
        }
        __syncthreads();
    }
}

size_t corb_chol_workspace_doubles(int n) { return (size_t)((n + CH_NB - 1) / CH_NB) * CH_NB * CH_NB; }
void corb_launch_chol_solve(double* A, int n, int ld, double* b, int* info, double* diag_ws, hipStream_t s)
{
    if (n <= 0) return;
    (void)hipMemsetAsync(info, 0, sizeof(int), s);
    const int np = (n + CH_NB - 1) / CH_NB;
    for (int k = 0; k < np; k++) {
        const int c0 = k * CH_NB;
        const int rows = n + 1 - c0;                                         // rows from the diagonal block on, the right-hand side row included
        hipLaunchKernelGGL(chol_panel_kernel, dim3((rows + CH_ROWS - 1) / CH_ROWS), dim3(64), 0, s, A, b, n, ld, k, info, diag_ws);
        const int trail = n - (c0 + CH_NB);
        if (trail > 0) {
            const int T = (trail + CH_ROWS - 1) / CH_ROWS;
            hipLaunchKernelGGL(chol_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, s, A, b, n, ld, k);
        }
    }
    hipLaunchKernelGGL(chol_backsub_kernel, dim3(1), dim3(1024), 0, s, A, b, n, ld, (const double*)diag_ws);
}
