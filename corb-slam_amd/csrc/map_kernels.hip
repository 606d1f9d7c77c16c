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

// the row of the N x N Hamming matrix with the least median (first on ties): MapPoint.cc:371-395.  One wavefront; dist = N shorts of LDS; D = N descriptors (4 x 64 bit)
__device__ __forceinline__ int distinctive_best(const unsigned long long* D, int N, unsigned short* dist, int lane)
{
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
    return best;
}

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
    const int best = distinctive_best(desc + (size_t)o0 * 4, N, dist_all[wave], lane);
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
// keyframe / map-point stores: record packing, staging and re-basing (device-to-device; see store_internal.h for the layouts)
#include "store_internal.h"
__global__ __launch_bounds__(256) void kf_pack_kernel(const CorbKeyPoint* kp, const uint8_t* desc, const float* ur, const float* depth, const int* count, int n_host,
                                                      unsigned long long id, char* rec, int F)
{
    const RecLayout L(F);
    const int n = min(n_host >= 0 ? n_host : *count, F);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        KfHeader* h = reinterpret_cast<KfHeader*>(rec);
        h->n = n; h->n_nodes = 0; h->m.id = id; h->m.client_id = 0; h->m.flags = 0; h->m.fx = h->m.fy = h->m.cx = h->m.cy = h->m.bf = 0.f; h->m.nlevels = 0; h->m.ba_global_for_kf = 0;
        for (int k = 0; k < 16; k++) { h->m.Tcw[k] = h->m.TcwGBA[k] = (k % 5 == 0) ? 1.f : 0.f; h->m.inv_level_sigma2[k] = 0.f; }
        *reinterpret_cast<int*>(rec + L.fv_off) = 0;
    }
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
    reinterpret_cast<unsigned long long*>(rec + L.mp_id)[i] = CORB_NO_MAP_POINT;
}
void corb_launch_kf_pack(const CorbKeyPoint* kp, const uint8_t* desc, const float* ur, const float* depth, const int* count, int n_host, unsigned long long id,
                         char* rec, int F, hipStream_t s)
{
    hipLaunchKernelGGL(kf_pack_kernel, dim3((F + 255) / 256), dim3(256), 0, s, kp, desc, ur, depth, count, n_host, id, rec, F);
}

// one thread per (record, 8 bytes of it): header words, then the observation arrays (entries past n_obs are cleared)
__global__ __launch_bounds__(256) void mp_pack_kernel(const CorbMapPointRecord* hdr, const int* obs_off, const unsigned long long* obs_kf, const uint32_t* obs_idx, int n,
                                                      char* base, int first, int O, int* status)
{
    const MpLayout L(O);
    const int i = blockIdx.x;                               // record
    if (i >= n) return;
    char* rec = base + (size_t)(first + i) * L.bytes;
    const int o0 = obs_off[i], cnt = obs_off[i + 1] - o0;
    if (cnt < 0 || cnt > O) { if (threadIdx.x == 0) atomicMax(status, 1); return; }
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(hdr + i);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(rec);
    for (int w = threadIdx.x; w < CORB_MP_HEADER_BYTES / 8; w += blockDim.x) dst[w] = w < (int)(sizeof(CorbMapPointRecord) / 8) ? src[w] : 0ull;
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<CorbMapPointRecord*>(rec)->n_obs = cnt;
    unsigned long long* ok = reinterpret_cast<unsigned long long*>(rec + L.obs_kf); uint32_t* oi = reinterpret_cast<uint32_t*>(rec + L.obs_idx);
    for (int k = threadIdx.x; k < O; k += blockDim.x) { ok[k] = k < cnt ? obs_kf[o0 + k] : 0ull; oi[k] = k < cnt ? obs_idx[o0 + k] : 0u; }
    unsigned long long* sc = reinterpret_cast<unsigned long long*>(rec + L.scratch);                  // CorbMapPointScratch: zero after a put, like the counters
    for (int w = threadIdx.x; w < (int)(sizeof(CorbMapPointScratch) / 8); w += blockDim.x) sc[w] = 0ull;
}
void corb_launch_mp_pack(const CorbMapPointRecord* hdr, const int* obs_off, const unsigned long long* obs_kf, const uint32_t* obs_idx, int n, char* base, int first, int O, int* status, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(mp_pack_kernel, dim3(n), dim3(64), 0, s, hdr, obs_off, obs_kf, obs_idx, n, base, first, O, status);
}
__global__ __launch_bounds__(64) void mp_unpack_kernel(const char* base, int first, int n, int O, CorbMapPointRecord* hdr, unsigned long long* obs_kf, uint32_t* obs_idx)
{
    const MpLayout L(O);
    const int i = blockIdx.x;
    if (i >= n) return;
    const char* rec = base + (size_t)(first + i) * L.bytes;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(rec);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(hdr + i);
    for (int w = threadIdx.x; w < (int)(sizeof(CorbMapPointRecord) / 8); w += blockDim.x) dst[w] = src[w];
    if (obs_kf) { const unsigned long long* ok = reinterpret_cast<const unsigned long long*>(rec + L.obs_kf); for (int k = threadIdx.x; k < O; k += blockDim.x) obs_kf[(size_t)i * O + k] = ok[k]; }
    if (obs_idx) { const uint32_t* oi = reinterpret_cast<const uint32_t*>(rec + L.obs_idx); for (int k = threadIdx.x; k < O; k += blockDim.x) obs_idx[(size_t)i * O + k] = oi[k]; }
}
void corb_launch_mp_unpack(const char* base, int first, int n, int O, CorbMapPointRecord* hdr, unsigned long long* obs_kf, uint32_t* obs_idx, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(mp_unpack_kernel, dim3(n), dim3(64), 0, s, base, first, n, O, hdr, obs_kf, obs_idx);
}

// staging of a push: dst[i] <- record slots[i]; records are multiples of 64 bytes, 16 bytes per lane
__global__ __launch_bounds__(256) void gather_records_kernel(const char* __restrict__ base, size_t rec_bytes, const int* __restrict__ slots, int n, char* __restrict__ dst)
{
    const size_t per = rec_bytes / 16;
    const size_t total = per * (size_t)n;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const size_t i = t / per, w = t - i * per;
        reinterpret_cast<uint4*>(dst + i * rec_bytes)[w] = reinterpret_cast<const uint4*>(base + (size_t)slots[i] * rec_bytes)[w];
    }
}
void corb_launch_gather_records(const char* base, size_t rec_bytes, const int* slots, int n, char* dst, hipStream_t s)
{
    if (n <= 0) return;
    const size_t total = rec_bytes / 16 * (size_t)n;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(gather_records_kernel, dim3(blocks), dim3(256), 0, s, base, rec_bytes, slots, n, dst);
}

// MapFusion::insertServerMapToGlobleMap on records (the arithmetic of rebase_map_kernel above, same roundings)
__global__ __launch_bounds__(256) void rebase_records_kernel(const float* __restrict__ To2n, char* kf_base, size_t kf_bytes, const int* kf_slots, int n_kf,
                                                             char* mp_base, size_t mp_bytes, const int* mp_slots, int n_mp)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float M[16];
#pragma unroll
    for (int k = 0; k < 16; k++) M[k] = To2n[k];
    if (i < n_kf) {
        float* T = reinterpret_cast<KfHeader*>(kf_base + (size_t)kf_slots[i] * kf_bytes)->m.Tcw; float t[16], o[16];
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
    if (i < n_mp) {
        float* p = reinterpret_cast<CorbMapPointRecord*>(mp_base + (size_t)mp_slots[i] * mp_bytes)->world_pos;
        const float d[3] = { __fsub_rn(p[0], M[3]), __fsub_rn(p[1], M[7]), __fsub_rn(p[2], M[11]) };
        float o[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            double s = __dmul_rn((double)M[r], (double)d[0]);
            s = __fma_rn((double)M[4 + r], (double)d[1], s);
            s = __fma_rn((double)M[8 + r], (double)d[2], s);
            o[r] = (float)s;
        }
        p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
    }
}
void corb_launch_rebase_records(const float* To2n, char* kf_base, size_t kf_bytes, const int* kf_slots, int n_kf, char* mp_base, size_t mp_bytes, const int* mp_slots, int n_mp, hipStream_t s)
{
    const int n = n_kf > n_mp ? n_kf : n_mp;
    if (n > 0) hipLaunchKernelGGL(rebase_records_kernel, dim3((n + 255) / 256), dim3(256), 0, s, To2n, kf_base, kf_bytes, kf_slots, n_kf, mp_base, mp_bytes, mp_slots, n_mp);
}

// one workgroup per keyframe: header, then every feature slot of the record (entries past the keyframe's count are cleared)
__global__ __launch_bounds__(256) void kf_pack_batch_kernel(const CorbKeyFrameMeta* meta, const int* feat_off, const CorbKeyPoint* kp, const uint8_t* desc, const float* ur,
                                                            const float* depth, const unsigned long long* mp_id, char* base, int first, int F)
{
    const RecLayout L(F);
    const int i = blockIdx.x;
    char* rec = base + (size_t)(first + i) * L.bytes;
    const int o0 = feat_off[i], n = min(feat_off[i + 1] - o0, F);
    if (threadIdx.x == 0) { KfHeader* h = reinterpret_cast<KfHeader*>(rec); h->n = n; h->n_nodes = 0; h->m = meta[i]; *reinterpret_cast<int*>(rec + L.fv_off) = 0; }
    for (int f = threadIdx.x; f < F; f += 256) {
        CorbKeyPoint k; k.x = k.y = k.size = k.response = 0.f; k.angle = 0.f; k.octave = 0; k.class_id = 0;
        float u = -1.f, dp = -1.f; unsigned long long id = CORB_NO_MAP_POINT;
        uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
        if (f < n) {
            k = kp[o0 + f];
            if (ur) u = ur[o0 + f];
            if (depth) dp = depth[o0 + f];
            if (mp_id) id = mp_id[o0 + f];
            if (desc) { const uint4* dsrc = reinterpret_cast<const uint4*>(desc + (size_t)(o0 + f) * 32); d0 = dsrc[0]; d1 = dsrc[1]; }
        }
        reinterpret_cast<CorbKeyPoint*>(rec + L.kp)[f] = k;
        uint4* dd = reinterpret_cast<uint4*>(rec + L.desc + (size_t)f * 32); dd[0] = d0; dd[1] = d1;
        reinterpret_cast<float*>(rec + L.ur)[f] = u; reinterpret_cast<float*>(rec + L.depth)[f] = dp;
        reinterpret_cast<float*>(rec + L.angle)[f] = k.angle;
        reinterpret_cast<uint8_t*>(rec + L.flags)[f] = 0;
        reinterpret_cast<unsigned long long*>(rec + L.mp_id)[f] = id;
    }
}
void corb_launch_kf_pack_batch(const CorbKeyFrameMeta* meta, const int* feat_off, const CorbKeyPoint* kp, const uint8_t* desc, const float* ur, const float* depth,
                               const unsigned long long* mp_id, int n, char* base, int first, int F, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(kf_pack_batch_kernel, dim3(n), dim3(256), 0, s, meta, feat_off, kp, desc, ur, depth, mp_id, base, first, F);
}

// ------------------------------------------------------------------------------------------------
// void MapPoint::Replace(MapPoint* pMP) (C/src/MapPoint.cc:277-316) on store records (corb_mp_store_replace).
// The observation lists are a handful of entries: one wavefront does the whole call -- lane 0 the sequential re-linking (the reference walks `obs` in map
// order and the outcome of an entry depends on the ones before it only through pMP's growing list), all lanes the descriptor distances of
// pMP->ComputeDistinctiveDescriptors().  Keyframes are found through an id table of the slots the caller names (built per call by kf_index_kernel).
#include "store_internal.h"
#include "device_util.h"
struct MpReplaceDev {
    char* mp_base; size_t mp_bytes; int max_obs; int slot_this, slot_into;
    char* kf_base; size_t kf_bytes; int F; CorbIdTable kfid;
    unsigned long long* desc;                                 // [max_obs][4] scratch: the descriptors pMP's observations contribute
    int* status;                                              // 0 done, 1 same point (no-op), CORB_ERR_CAPACITY: pMP's list has no room (nothing written)
};
__global__ __launch_bounds__(256) void kf_index_kernel(const char* kf_base, size_t kf_bytes, int first, int n, CorbIdTable idt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const KfHeader* h = reinterpret_cast<const KfHeader*>(kf_base + (size_t)(first + i) * kf_bytes);
    // a slot without features cannot hold an observation: never-filled slots (zero-initialised: id 0, n 0) stay out of the table, so a real keyframe with mnId 0
    // is not shadowed by them; of two filled slots with one id the lower slot is the keyframe (the table's vals are preset for atomicMin)
    if (h->n <= 0) return;
    corb_idtab_insert_min(idt, h->m.id, first + i);
}
__global__ __launch_bounds__(64) void mp_replace_kernel(MpReplaceDev t)
{
    __shared__ unsigned short dist[DD_MAX_OBS];
    __shared__ int sh_n;
    const int lane = threadIdx.x;
    const MpLayout L(t.max_obs);
    char* rthis = t.mp_base + (size_t)t.slot_this * t.mp_bytes; char* rinto = t.mp_base + (size_t)t.slot_into * t.mp_bytes;
    CorbMapPointRecord* a = reinterpret_cast<CorbMapPointRecord*>(rthis); CorbMapPointRecord* b = reinterpret_cast<CorbMapPointRecord*>(rinto);
    CorbMapPointCounters* ca = reinterpret_cast<CorbMapPointCounters*>(rthis + sizeof(CorbMapPointRecord)); CorbMapPointCounters* cb = reinterpret_cast<CorbMapPointCounters*>(rinto + sizeof(CorbMapPointRecord));
    unsigned long long* akf = reinterpret_cast<unsigned long long*>(rthis + L.obs_kf); uint32_t* aidx = reinterpret_cast<uint32_t*>(rthis + L.obs_idx);
    unsigned long long* bkf = reinterpret_cast<unsigned long long*>(rinto + L.obs_kf); uint32_t* bidx = reinterpret_cast<uint32_t*>(rinto + L.obs_idx);
    const RecLayout KL(t.F);
    if (lane == 0) {
        sh_n = -1;
        if (a->id == b->id) *t.status = 1;                                                   // if (pMP->mnId == this->mnId) return;
        else {
            const int na = min(a->n_obs, t.max_obs); int nb = min(b->n_obs, t.max_obs);
            int fresh = 0;
            for (int k = 0; k < na; k++) { bool in = false; for (int j = 0; j < nb; j++) in = in || bkf[j] == akf[k]; fresh += in ? 0 : 1; }
            if (nb + fresh > t.max_obs) *t.status = CORB_ERR_CAPACITY;
            else {
                for (int k = 0; k < na; k++) {                                               // obs in map order (:300-313)
                    const unsigned long long kid = akf[k]; const uint32_t idx = aidx[k];
                    bool in = false; for (int j = 0; j < nb; j++) in = in || bkf[j] == kid;  // pMP->IsInKeyFrame(pKF)
                    const int ks = corb_idtab_find(t.kfid, kid);
                    unsigned long long* mp_id = ks >= 0 ? reinterpret_cast<unsigned long long*>(t.kf_base + (size_t)ks * t.kf_bytes + KL.mp_id) : nullptr;
                    if (!in) {
                        if (mp_id && (int)idx < t.F) mp_id[idx] = b->id;                     // pKF->ReplaceMapPointMatch(mit->second, pMP)
                        int j = nb;                                                          // pMP->AddObservation(pKF, mit->second): the list ascends in the keyframe id
                        while (j > 0 && bkf[j - 1] > kid) { bkf[j] = bkf[j - 1]; bidx[j] = bidx[j - 1]; j--; }
                        bkf[j] = kid; bidx[j] = idx; nb++;
                    } else if (mp_id && (int)idx < t.F) mp_id[idx] = CORB_NO_MAP_POINT;      // pKF->EraseMapPointMatch(mit->second)
                }
                b->n_obs = nb;
                cb->n_found += ca->n_found; cb->n_visible += ca->n_visible;                 // pMP->IncreaseFound(nfound); pMP->IncreaseVisible(nvisible)
                a->n_obs = 0; a->flags |= CORB_MP_BAD; ca->replaced_by = b->id + 1ull;       // mObservations.clear(); mbBad = true; mpReplaced = pMP
                for (int k = 0; k < t.max_obs; k++) { akf[k] = 0ull; aidx[k] = 0u; }
                *t.status = 0;
                // pMP->ComputeDistinctiveDescriptors() (:337-402): the descriptors of its observations in non-bad keyframes, in list order
                int m = 0;
                if (!(b->flags & CORB_MP_BAD)) {
                    for (int j = 0; j < nb; j++) {
                        const int ks = corb_idtab_find(t.kfid, bkf[j]);
                        if (ks < 0) continue;
                        const char* kr = t.kf_base + (size_t)ks * t.kf_bytes;
                        if ((reinterpret_cast<const KfHeader*>(kr)->m.flags & CORB_KF_BAD) || (int)bidx[j] >= t.F) continue;      // if(!pKF->isBad())
                        const unsigned long long* d = reinterpret_cast<const unsigned long long*>(kr + KL.desc) + 4 * (size_t)bidx[j];
                        t.desc[4 * m] = d[0]; t.desc[4 * m + 1] = d[1]; t.desc[4 * m + 2] = d[2]; t.desc[4 * m + 3] = d[3]; m++;
                    }
                }
                sh_n = m;
            }
        }
    }
    __syncthreads();
    const int m = sh_n;
    if (m <= 0 || m > DD_MAX_OBS) return;                                                    // (if(vDescriptors.empty()) return;)
    __threadfence_block();
    const int best = distinctive_best(t.desc, m, dist, lane);
    if (lane < 4) reinterpret_cast<unsigned long long*>(b->descriptor)[lane] = t.desc[4 * best + lane];       // mDescriptor = vDescriptors[BestIdx].clone()
}
void corb_launch_mp_replace(char* mp_base, size_t mp_bytes, int max_obs, int slot_this, int slot_into, char* kf_base, size_t kf_bytes, int F, int kf_first, int kf_n,
                            CorbIdTable kfid, unsigned long long* desc, int* status, hipStream_t s)
{
    if (kf_n > 0) hipLaunchKernelGGL(kf_index_kernel, dim3((kf_n + 255) / 256), dim3(256), 0, s, kf_base, kf_bytes, kf_first, kf_n, kfid);
    MpReplaceDev t; t.mp_base = mp_base; t.mp_bytes = mp_bytes; t.max_obs = max_obs; t.slot_this = slot_this; t.slot_into = slot_into;
    t.kf_base = kf_base; t.kf_bytes = kf_bytes; t.F = F; t.kfid = kfid; t.desc = desc; t.status = status;
    hipLaunchKernelGGL(mp_replace_kernel, dim3(1), dim3(64), 0, s, t);
}
// the 16 spare bytes of the record headers <-> CorbMapPointCounters
__global__ __launch_bounds__(256) void mp_counters_kernel(char* base, size_t bytes, int first, int n, CorbMapPointCounters* io, int set)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    CorbMapPointCounters* c = reinterpret_cast<CorbMapPointCounters*>(base + (size_t)(first + i) * bytes + sizeof(CorbMapPointRecord));
    if (set) *c = io[i]; else io[i] = *c;
}
// CorbMapPointScratch (behind the observation lists) <-> an array; 13 eight-byte words per record, a thread per word
__global__ __launch_bounds__(256) void mp_scratch_kernel(char* base, size_t bytes, size_t off, int first, int n, unsigned long long* io, int set)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    constexpr int W = (int)(sizeof(CorbMapPointScratch) / 8);
    if (t >= (size_t)n * W) return;
    const size_t i = t / W; const int w = (int)(t - i * W);
    unsigned long long* r = reinterpret_cast<unsigned long long*>(base + (size_t)(first + i) * bytes + off) + w;
    if (set) *r = io[t]; else io[t] = *r;
}
void corb_launch_mp_scratch(char* base, size_t bytes, size_t off, int first, int n, void* io, int set, hipStream_t s)
{
    const size_t words = (size_t)n * (sizeof(CorbMapPointScratch) / 8);
    if (n > 0) hipLaunchKernelGGL(mp_scratch_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, base, bytes, off, first, n, (unsigned long long*)io, set);
}
void corb_launch_mp_counters(char* base, size_t bytes, int first, int n, CorbMapPointCounters* io, int set, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(mp_counters_kernel, dim3((n + 255) / 256), dim3(256), 0, s, base, bytes, first, n, io, set);
}
