// map_kernels.hip -- map maintenance arithmetic next to the hot path:
//   MapPoint::ComputeDistinctiveDescriptors (C/src/MapPoint.cc:337-402): per map point N x N Hamming distances, median of every
//     row, first row with the least median -- one wavefront per map point, row distances in LDS, the median by a 9-bit
//     radix select with wave ballots (distances are 0..256);
//   MapFusion::insertServerMapToGlobleMap (S/src/MapFusion.cpp:622-658): rigid re-basing of a client's sub-map into the
//     global map, Tcw <- Tcw * To2n, p <- Rwc (p - tcw) (cv::gemm: double accumulation, one rounding).
#include "corb_internal.h"

#define DD_MAX_OBS 1024        // observations per map point held in LDS (more is reported, never truncated silently)

__device__ __forceinline__ int mk_hamming(const unsigned long long* a, const unsigned long long* b)
{ return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]); }

__global__ __launch_bounds__(256) void distinctive_desc_kernel(const unsigned long long* __restrict__ desc, const int* __restrict__ offset, int n_points,
                                                               int* __restrict__ best_idx, int* __restrict__ status)
{
    __shared__ unsigned short dist_all[4][DD_MAX_OBS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wave;
    if (p >= n_points) return;
    const int o0 = offset[p], N = offset[p + 1] - o0;
    if (N <= 0) { if (lane == 0) best_idx[p] = -1; return; }
    if (N > DD_MAX_OBS) { if (lane == 0) { best_idx[p] = -1; *status = CORB_ERR_OVERFLOW; } return; }
    unsigned short* dist = dist_all[wave];
    const unsigned long long* D = desc + (size_t)o0 * 4;
    const int kth = (int)(0.5 * (double)(N - 1));                   // vDists[0.5*(N-1)]
    int best_median = 0x7FFFFFFF, best = 0;
    for (int i = 0; i < N; i++) {
        const unsigned long long a[4] = { D[(size_t)i * 4], D[(size_t)i * 4 + 1], D[(size_t)i * 4 + 2], D[(size_t)i * 4 + 3] };
        for (int j = lane; j < N; j += 64) dist[j] = (unsigned short)(j == i ? 0 : mk_hamming(a, D + (size_t)j * 4));
        // k-th smallest of dist[0..N): fix the bits from the top; `cand` = elements agreeing with the prefix so far
        int prefix = 0, k = kth;
        for (int bit = 8; bit >= 0; bit--) {
            int zeros = 0;
            for (int j0 = 0; j0 < N; j0 += 64) {
                const int j = j0 + lane;
                const bool is_zero = j < N && ((dist[j] >> (bit + 1)) == (prefix >> (bit + 1))) && !((dist[j] >> bit) & 1);
                zeros += __popcll(__ballot(is_zero));
            }
            if (k >= zeros) { k -= zeros; prefix |= 1 << bit; }
        }
        if (prefix < best_median) { best_median = prefix; best = i; }
    }
    if (lane == 0) best_idx[p] = best;
}

__global__ __launch_bounds__(256) void rebase_map_kernel(const float* __restrict__ To2n, float* __restrict__ poses, int n_poses, float* __restrict__ points, int n_points)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float M[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M[k] = To2n[k];
    if (i < n_poses) {
        float* T = poses + 16 * (size_t)i; float t[16], o[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = T[k];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                double s = __dmul_rn((double)t[r * 4], (double)M[c]);
                s = __fma_rn((double)t[r * 4 + 1], (double)M[4 + c], s);
                s = __fma_rn((double)t[r * 4 + 2], (double)M[8 + c], s);
                s = __fma_rn((double)t[r * 4 + 3], (double)M[12 + c], s);
                o[r * 4 + c] = (float)s;
            }
#pragma unroll
        for (int k = 0; k < 16; k++) T[k] = o[k];
    }
    if (i < n_points) {
        float* p = points + 3 * (size_t)i;
        const float d[3] = { __fsub_rn(p[0], M[3]), __fsub_rn(p[1], M[7]), __fsub_rn(p[2], M[11]) };
#pragma unroll
        for (int r = 0; r < 3; r++) {
            double s = __dmul_rn((double)M[r], (double)d[0]);
            s = __fma_rn((double)M[4 + r], (double)d[1], s);
            s = __fma_rn((double)M[8 + r], (double)d[2], s);
            p[r] = (float)s;
        }
    }
}

void corb_launch_distinctive(const unsigned long long* desc, const int* offset, int n_points, int* best_idx, int* status, hipStream_t s)
{
    if (n_points > 0) hipLaunchKernelGGL(distinctive_desc_kernel, dim3((n_points + 3) / 4), dim3(256), 0, s, desc, offset, n_points, best_idx, status);
}
void corb_launch_rebase(const float* To2n, float* poses, int n_poses, float* points, int n_points, hipStream_t s)
{
    const int n = n_poses > n_points ? n_poses : n_points;
    if (n > 0) hipLaunchKernelGGL(rebase_map_kernel, dim3((n + 255) / 256), dim3(256), 0, s, To2n, poses, n_poses, points, n_points);
}

// ------------------------------------------------------------------------------------------------
// keyframe store: one slot record <- a keyframe's results (device-to-device; see store_internal.h for the layout)
#include "store_internal.h"
__global__ __launch_bounds__(256) void kf_pack_kernel(const CorbKeyPoint* kp, const uint8_t* desc, const float* ur, const float* depth, const int* count, int n_host,
                                                      unsigned long long id, char* rec, int F)
{
    const RecLayout L(F);
    const int n = min(n_host >= 0 ? n_host : *count, F);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { int* h = reinterpret_cast<int*>(rec); h[0] = n; h[1] = 0; *reinterpret_cast<unsigned long long*>(rec + 8) = id; *reinterpret_cast<int*>(rec + L.fv_off) = 0; }
    if (i >= F) return;
    CorbKeyPoint k; k.x = k.y = k.size = k.response = 0.f; k.angle = 0.f; k.octave = 0; k.class_id = 0;
    float u = -1.f, dp = -1.f;
    uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
    if (i < n) {
        k = kp[i]; u = ur[i]; dp = depth[i];
        const uint4* dsrc = reinterpret_cast<const uint4*>(desc + (size_t)i * 32); d0 = dsrc[0]; d1 = dsrc[1];
    }
    reinterpret_cast<CorbKeyPoint*>(rec + L.kp)[i] = k;
    uint4* dd = reinterpret_cast<uint4*>(rec + L.desc + (size_t)i * 32); dd[0] = d0; dd[1] = d1;
    reinterpret_cast<float*>(rec + L.ur)[i] = u; reinterpret_cast<float*>(rec + L.depth)[i] = dp;
    reinterpret_cast<float*>(rec + L.angle)[i] = k.angle;
    reinterpret_cast<uint8_t*>(rec + L.flags)[i] = 0;
}
void corb_launch_kf_pack(const CorbKeyPoint* kp, const uint8_t* desc, const float* ur, const float* depth, const int* count, int n_host, unsigned long long id,
                         char* rec, int F, hipStream_t s)
{
    hipLaunchKernelGGL(kf_pack_kernel, dim3((F + 255) / 256), dim3(256), 0, s, kp, desc, ur, depth, count, n_host, id, rec, F);
}
