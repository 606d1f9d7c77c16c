// stream_patterns.hip -- what does the access pattern of the BA solve's two CG kernels cost against a plain coalesced stream?  (development aid, gfx950)
//   0  coalesced: lane l of a wavefront reads 16 B at 16 l, a wavefront covers 1 KB per load, 2 880 B per trip like pattern 1
//   1  spmv-like: 60 lanes = 10 groups x 6 rows; a lane reads its 48-byte block row (3 x 16 B at stride 48 across the lanes), 3 trips per wavefront
//   2  pattern 1 + the dependent column index load and the two 48-byte vector gathers (the real kernel's chain)
//   3  pattern 0 + the same dependent index / gathers
//   4  step_big-like: fp32, 16 lanes per row read 64 B, 4 rows (stride 384 B) per wavefront load, 6 loads per pass, 3 passes
//   5  step_big coalesced: fp32, lane reads 16 B at 16 l, 18 loads per wavefront
// build: hipcc --offload-arch=gfx950 -O3 -w stream_patterns.hip -o stream_patterns ; run: gpurun -- tools/ubench/stream_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>

#define NROW 50000
#define BPR 30                          // blocks per block row
//   6  pattern 2 behind the real kernel's head: two flags and three scalars loaded, a convergence test, beta
//   7  pattern 6 + the tail's arithmetic: 10 shuffles, the 6 result lanes store p and q, workgroup sum of p.q (two barriers), thread 0 stores the partial
//   8  pattern 7 + the device-scope part: atomic publish, wait, group ticket (wavefront 0 only), the last taker sums the group
__global__ __launch_bounds__(256) void k_spmv(const double* __restrict__ val, const int* __restrict__ col, const double* __restrict__ x, double* out, int mode,
                                              const double* scal, const int* flag, double* part, int* tick, double* vec, const int* rowptr)
{
    __shared__ double red[4];
    double beta = 0.5;
    const int m0 = mode;
    mode &= 15;
    if (mode >= 6) {
        if (flag[0] || flag[1]) return;
        const double rr = scal[0], rz_new = scal[1], rz_old = scal[2];
        if (rr <= 1e-16 * scal[3]) { if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = rr; return; }
        beta = rz_new / rz_old;
        mode = 2;
    }
    const int lane = threadIdx.x & 63, k = blockDim.x == 64 ? (int)blockIdx.x : blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= NROW) return;
    double q = 0;
    if (mode == 1 || mode == 2) {
        const int grp = lane / 6, a = lane - 6 * grp;
        const int ku = (m0 & 16) ? __builtin_amdgcn_readfirstlane(k) : k;       // mode + 16: the row index as a scalar (row pointers through the scalar cache)
        if (lane < 60)
            for (int s = (rowptr ? rowptr[ku] : k * BPR) + grp, se = rowptr ? rowptr[ku + 1] : (k + 1) * BPR; s < se; s += 10) {
                const double* Sv = val + (size_t)s * 36 + a * 6;
                if (mode == 2) {
                    const int j = col[s];
                    const double* xj = x + 6 * (size_t)j; const double* yj = x + 6 * (size_t)NROW + 6 * (size_t)j;
#pragma unroll
                    for (int c = 0; c < 6; c++) q += Sv[c] * (xj[c] + beta * yj[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < 6; c++) q += Sv[c];
                }
            }
    } else {
        // the row's 30 blocks = 8 640 B = 540 x 16 B: 9 loads of 60 lanes x 16 B, 3 per trip
        const double2* V = reinterpret_cast<const double2*>(val + (size_t)k * BPR * 36);
        if (lane < 60)
            for (int t = 0; t < 3; t++) {
                if (mode == 3) {
                    const int s = k * BPR + t * 10 + lane / 6;
                    const int j = col[s];
                    const double* xj = x + 6 * (size_t)j; const double* yj = x + 6 * (size_t)NROW + 6 * (size_t)j;
#pragma unroll
                    for (int u = 0; u < 3; u++) { const double2 v = V[(t * 3 + u) * 60 + lane]; q += v.x * (xj[2 * u] + 0.5 * yj[2 * u]) + v.y * (xj[2 * u + 1] + 0.5 * yj[2 * u + 1]); }
                } else {
#pragma unroll
                    for (int u = 0; u < 3; u++) { const double2 v = V[(t * 3 + u) * 60 + lane]; q += v.x + v.y; }
                }
            }
    }
    if ((m0 & 15) >= 7) {
        double qt = 0;
#pragma unroll
        for (int m = 0; m < 10; m++) qt += __shfl(q, (lane % 6) + 6 * m);
        double pq = 0;
        if (lane < 6) { const size_t i = 6 * (size_t)k + lane; const double pi = x[i] + beta * x[6 * (size_t)NROW + i]; vec[i] = pi; vec[6 * (size_t)NROW + i] = qt; pq = pi * qt; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pq += __shfl_xor(pq, o);
        __syncthreads();
        if (lane == 0) red[threadIdx.x >> 6] = pq;
        __syncthreads();
        const double s1 = red[0] + red[1] + red[2] + red[3];
        if ((m0 & 15) == 7) { if (threadIdx.x == 0) part[blockIdx.x] = s1; return; }
        if (threadIdx.x == 0) __hip_atomic_store(&part[blockIdx.x], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x >= 64) return;
        const int grp = blockIdx.x / 64, first = grp * 64, n_in = min(64, (int)gridDim.x - first);
        int last = 0;
        if (lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = __hip_atomic_fetch_add(tick + (size_t)grp * 64, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_in - 1;
        }
        if (!__shfl(last, 0)) return;
        double v0 = lane < n_in ? __hip_atomic_load(part + first + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v0 += __shfl_xor(v0, o);
        if (lane == 0) { tick[(size_t)grp * 64] = 0; part[100000 + grp] = v0; }
        return;
    }
    if (q == 1.2345e300) out[k] = q;
}

#define NBLK 3125                       // preconditioner blocks of 96 x 96 floats
__global__ void k_fill(double* v, size_t n) { for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v[i] = 1.0 + (double)((i * 2654435761ull) & 0xFFFFF) * 1e-6; }
__global__ __launch_bounds__(256) void k_pc(const float* __restrict__ pc, double* out, int mode)
{
    __shared__ double rn[96];
    const int b = blockIdx.x >> 1, slice = blockIdx.x & 1;
    if (threadIdx.x < 96) rn[threadIdx.x] = 1.0 + threadIdx.x;
    __syncthreads();
    const float* D = pc + (size_t)b * 96 * 96;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double acc = 0;
    if (mode == 4) {
        const int l16 = lane & 15, rsub = lane >> 4;
#pragma unroll
        for (int pass = 0; pass < 3; pass++) {
            const int t = slice * 48 + pass * 16 + wave * 4 + rsub;
            const float* Dr = D + (size_t)t * 96;
#pragma unroll 6
            for (int c = l16; c < 96; c += 16) acc += (double)Dr[c] * rn[c];
        }
    } else {
        // the slice's 48 rows = 18 432 B = 1 152 x 16 B: 4.5 loads of 256 threads
        const float4* V = reinterpret_cast<const float4*>(D + (size_t)slice * 48 * 96);
        for (int i = threadIdx.x; i < 1152; i += 256) {
            const float4 v = V[i]; const int c = (i % 24) * 4;
            acc += (double)v.x * rn[c] + (double)v.y * rn[c + 1] + (double)v.z * rn[c + 2] + (double)v.w * rn[c + 3];
        }
    }
    if (acc == 1.2345e300) out[blockIdx.x] = acc;
}

//   9  store 144-byte blocks, one per lane (9 x 16 B at a lane stride of 144 B: how the BA kernels write V and JB | r)
//  10  the same bytes staged through LDS and stored with consecutive lanes on consecutive 16 B (full lines per instruction)
__global__ __launch_bounds__(256) void k_store(double2* out, size_t nblk, int mode)
{
    __shared__ double2 stage[4][64 * 9];
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= nblk) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double2 v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = make_double2((double)i + k, (double)i - k);
    if (mode == 9) {
#pragma unroll
        for (int k = 0; k < 9; k++) out[i * 9 + k] = v[k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) stage[w][lane * 9 + k] = v[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double2* o = out + (i - lane) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k * 64 + lane] = stage[w][k * 64 + lane];
    }
}

int main()
{
    const size_t nval = (size_t)NROW * BPR * 36;
    double *val, *x, *out; int* col; float* pc;
    hipMalloc(&val, 2 * nval * 8); hipMalloc(&x, 12 * (size_t)NROW * 8); hipMalloc(&out, 8 * (size_t)NROW); hipMalloc(&col, 4 * (size_t)NROW * BPR);
    hipMalloc(&pc, 5 * (size_t)NBLK * 96 * 96 * 4);
    hipMemset(val, 0, 2 * nval * 8); hipMemset(x, 0, 12 * (size_t)NROW * 8); hipMemset(pc, 0, 5 * (size_t)NBLK * 96 * 96 * 4);
    std::vector<int> hc((size_t)NROW * BPR);
    srand(1);
    for (int k = 0; k < NROW; k++) for (int s = 0; s < BPR; s++) { long j = k + (s - 15) * (s % 3 == 0 ? 37 : 1); if (j < 0) j = 0; if (j >= NROW) j = NROW - 1; hc[(size_t)k * BPR + s] = (int)j; }
    hipMemcpy(col, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    double* scal; int* flag; double* part; int* tick; double* vec;
    hipMalloc(&scal, 64); hipMalloc(&flag, 64); hipMalloc(&part, 8 * 131072); hipMalloc(&tick, 4 * 64 * 1024); hipMalloc(&vec, 12 * (size_t)NROW * 8);
    { double hs[4] = { 1.0, 1.0, 2.0, 1.0 }; hipMemcpy(scal, hs, 32, hipMemcpyHostToDevice); hipMemset(flag, 0, 64); hipMemset(tick, 0, 4 * 64 * 1024); }
    int* rowptr; hipMalloc(&rowptr, 4 * (size_t)(NROW + 1));
    { std::vector<int> rp(NROW + 1); rp[0] = 0; long tot = 0;
      std::vector<int> lens(NROW);
      for (int k = 0; k < NROW; k++) { int len = 6 + rand() % 49; if (rand() % 50 == 0 && !getenv("NOLONG")) len = 120 + rand() % 100; if (getenv("MULT10")) len = ((len + 5) / 10) * 10; if (len < 10) len = 10; lens[k] = len; }
      if (getenv("SORTED")) std::sort(lens.begin(), lens.end(), [](int a, int b) { return a > b; });
      for (int k = 0; k < NROW; k++) { tot += lens[k]; rp[k + 1] = (int)tot; }
      // scale to the same total number of blocks
      for (int k = 0; k <= NROW; k++) rp[k] = (int)((double)rp[k] * ((double)NROW * BPR / (double)tot));
      if (getenv("EQ")) for (int k = 0; k <= NROW; k++) rp[k] = k * BPR;
      hipMemcpy(rowptr, rp.data(), 4 * (size_t)(NROW + 1), hipMemcpyHostToDevice); }
    const bool real_like = getenv("REAL") != nullptr;
    if (real_like) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, val, 2 * nval); hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, x, 12 * (size_t)NROW); hipDeviceSynchronize(); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mi = 0; mi <= 10; mi++) { const int mode = mi <= 8 ? mi : (mi == 9 ? 18 : 22);
        const bool pcm = mode == 4 || mode == 5;
        if (mi > 8 && !real_like) continue;
        const double bytes = pcm ? (double)NBLK * 96 * 96 * 4 : (double)nval * 8;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            for (int it = 0; it < 50; it++) {
                if (pcm) hipLaunchKernelGGL(k_pc, dim3(NBLK * 2), dim3(256), 0, 0, pc + (size_t)(it % 5) * NBLK * 96 * 96, out, mode);   // (rotating buffers: the 256 MB Infinity Cache must not serve the stream)
                else if (getenv("WG64")) hipLaunchKernelGGL(k_spmv, dim3(NROW), dim3(64), 0, 0, val + (size_t)(it % 2) * nval, col, x, out, mode, scal, flag, part, tick, vec, real_like ? rowptr : nullptr);
                else hipLaunchKernelGGL(k_spmv, dim3((NROW + 3) / 4), dim3(256), 0, 0, val + (size_t)(it % 2) * nval, col, x, out, mode, scal, flag, part, tick, vec, real_like ? rowptr : nullptr);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d: %.1f us per launch, %.0f MB, %.2f TB/s\n", mode, ms * 1e3 / 50, bytes / 1e6, bytes / (ms * 1e-3 / 50) / 1e12);
        }
    }
    {
        const size_t nb = 27500000 / 64 * 64;
        double2* vout; hipMalloc(&vout, nb * 144);
        for (int mode = 9; mode <= 10; mode++)
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                for (int it = 0; it < 5; it++) hipLaunchKernelGGL(k_store, dim3((unsigned)(nb / 256)), dim3(256), 0, 0, vout, nb, mode);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("mode %d: %.1f us per launch, %.0f MB written, %.2f TB/s\n", mode, ms * 1e3 / 5, nb * 144.0 / 1e6, nb * 144.0 / (ms * 1e-3 / 5) / 1e12);
            }
    }
    return 0;
}
