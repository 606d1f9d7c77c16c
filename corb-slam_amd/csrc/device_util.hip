// device_util.hip -- device-wide exclusive scan of 32-bit counts (three launches: per-tile sums, scan of the tile sums by one workgroup, per-tile scan + offset)
#include "device_util.h"
#include "lane_exchange.h"

#define SCAN_T 256
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_T * SCAN_ITEMS)

size_t corb_scan_scratch_ints(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 2; }

__device__ __forceinline__ int scan_block_sum(int v, int* sh)      // sum over the workgroup, returned to every thread
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < SCAN_T / 64; i++) t += sh[i];
    __syncthreads();
    return t;
}
// exclusive prefix of v over the workgroup (thread order)
__device__ __forceinline__ int scan_block_excl(int v, int* sh)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
    inc = lx_wave_incl_scan_i(inc);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; i++) base += sh[i];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_T) void scan_tile_sums_kernel(const int* __restrict__ in, size_t n, int* __restrict__ tile_sum)
{
    __shared__ int sh[SCAN_T / 64];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
    int v = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) { const size_t i = base + (size_t)k * SCAN_T + threadIdx.x; if (i < n) v += in[i]; }
    const int t = scan_block_sum(v, sh);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = t;
}
// one workgroup: tile_sum[0..nt) -> exclusive prefixes in place, the grand total in tile_sum[nt]
__global__ __launch_bounds__(SCAN_T) void scan_tile_prefix_kernel(int* tile_sum, size_t nt)
{
    __shared__ int sh[SCAN_T / 64];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t b = 0; b < nt; b += SCAN_T) {
        const size_t i = b + threadIdx.x;
        const int v = i < nt ? tile_sum[i] : 0;
        const int ex = scan_block_excl(v, sh);
        const int carry = carry_s;
        if (i < nt) tile_sum[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == SCAN_T - 1) carry_s = carry + ex + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sum[nt] = carry_s;
}
__global__ __launch_bounds__(SCAN_T) void scan_apply_kernel(const int* in, int* out, size_t n, const int* __restrict__ tile_sum, size_t nt)
{
    __shared__ int sh[SCAN_T / 64];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;      // a thread owns SCAN_ITEMS consecutive entries
    int v[SCAN_ITEMS]; int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
    int run = tile_sum[blockIdx.x] + scan_block_excl(s, sh);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tile_sum[nt];
}

// Up to SCAN1_MAX counts: ONE workgroup, a thread owning ceil(n / 1024) consecutive entries -- a local window's scans (a few thousand map points, a few dozen keyframes)
// are dependent launches of a few microseconds each, and three per scan were most of the graph's set-up time.  Same integers as the three-launch form; in == out allowed.
#define SCAN1_T 1024
#define SCAN1_MAX 32768
__global__ __launch_bounds__(SCAN1_T) void scan_one_wg_kernel(const int* in, int* out, int n)
{
    __shared__ int sh[SCAN1_T / 64];
    const int per = (n + SCAN1_T - 1) / SCAN1_T;
    const int b = min((int)threadIdx.x * per, n), e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; i++) sum += in[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = sum;
    inc = lx_wave_incl_scan_i(inc);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int i = 0; i < SCAN1_T / 64; i++) { if (i < w) base += sh[i]; total += sh[i]; }
    int run = base + inc - sum;
    for (int i = b; i < e; i++) { const int v = in[i]; out[i] = run; run += v; }
    if (threadIdx.x == 0) out[n] = total;
}

// up to four independent scans of at most SCAN1_MAX counts each as ONE launch, a workgroup per array (the flattening of a local window scans four arrays back to back)
struct Scan4 { const int* in[4]; int* out[4]; int n[4]; };
__global__ __launch_bounds__(SCAN1_T) void scan4_one_wg_kernel(Scan4 a)
{
    __shared__ int sh[SCAN1_T / 64];
    const int* in = a.in[blockIdx.x]; int* out = a.out[blockIdx.x]; const int n = a.n[blockIdx.x];
    const int per = (n + SCAN1_T - 1) / SCAN1_T;
    const int b = min((int)threadIdx.x * per, n), e = min(b + per, n);
    int sum = 0;
    for (int i = b; i < e; i++) sum += in[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = sum;
    inc = lx_wave_incl_scan_i(inc);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int i = 0; i < SCAN1_T / 64; i++) { if (i < w) base += sh[i]; total += sh[i]; }
    int run = base + inc - sum;
    for (int i = b; i < e; i++) { const int v = in[i]; out[i] = run; run += v; }
    if (threadIdx.x == 0) out[n] = total;
}
bool corb_launch_exclusive_scan4(const int* const* in, int* const* out, const size_t* n, int count, hipStream_t s)
{
    if (count < 1 || count > 4) return false;
    Scan4 a;
    for (int k = 0; k < count; k++) { if (n[k] > SCAN1_MAX) return false; a.in[k] = in[k]; a.out[k] = out[k]; a.n[k] = (int)n[k]; }
    hipLaunchKernelGGL(scan4_one_wg_kernel, dim3(count), dim3(SCAN1_T), 0, s, a);
    return true;
}

void corb_launch_exclusive_scan(const int* in, int* out, size_t n, int* scratch, hipStream_t s)
{
    const size_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nt == 0) { (void)hipMemsetAsync(out, 0, sizeof(int), s); return; }
    if (n <= SCAN1_MAX) { hipLaunchKernelGGL(scan_one_wg_kernel, dim3(1), dim3(SCAN1_T), 0, s, in, out, (int)n); return; }
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nt), dim3(SCAN_T), 0, s, in, n, scratch);
    hipLaunchKernelGGL(scan_tile_prefix_kernel, dim3(1), dim3(SCAN_T), 0, s, scratch, nt);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nt), dim3(SCAN_T), 0, s, in, out, n, scratch, nt);
}
