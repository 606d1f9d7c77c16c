// match_kernels.hip -- 256-bit Hamming matching on gfx950 wavefronts.
//   stereo_match_kernel / stereo_filter_kernel : Frame::ComputeStereoMatches (C/src/Frame.cc:470-644)
//   bow_match_kernel / bow_finalize_kernel     : ORBmatcher::SearchByBoW x3 (C/src/ORBmatcher.cc:162-423, 657-790)
//   tri_match_kernel                           : ORBmatcher::SearchForTriangulation (C/src/ORBmatcher.cc:792-958)
// A descriptor is 4 x u64; distance = 4 x __popcll (== the SWAR popcount of ORBmatcher.cc:1792-1808).
// One wavefront owns one query; candidates are strided over the 64 lanes; best / second-best are
// merged with wave shuffles, keys are (distance << 16 | order) so the reference's first/last-wins
// tie rules are reproduced exactly.
#include "corb_internal.h"
#include "match_internal.h"
#include <limits.h>
#include "lane_exchange.h"

// (wave reductions through v_permlane swaps and DPP moves: lane_exchange.h)
__device__ __forceinline__ int wmin_i32(int v) { return lx_wave_min_i(v); }
__device__ __forceinline__ unsigned wmin_u32(unsigned v) { return lx_wave_min_u(v); }
__device__ __forceinline__ int wsum_i32(int v) { return lx_wave_sum_i(v); }
__device__ __forceinline__ int hamming256(const unsigned long long* a, const unsigned long long* b) {
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// ------------------------------------------------------------------------------------------------
// Row table of Frame::ComputeStereoMatches (C/src/Frame.cc:481-497): right keypoint iR is a candidate
// on every row floor(y-r)..ceil(y+r), r = 2*scale[octave].  CSR per frame, built by one workgroup:
// count (LDS atomics) -> scan -> fill.  Order inside a row is irrelevant: the matcher keeps the
// minimum (distance << 16 | iR), which is the reference's "first iR with the smallest distance".
#define SR_T 256             // one workgroup per frame is a pure latency chain.  Alone, 1 024 threads are faster (20 vs 28 us per 256 frames: more LDS atomics / loads
                             // in flight), but a 16-wavefront workgroup waits for 16 free wave slots on ONE CU while the other part-batch's kernels keep refilling them:
                             // in the pipeline 256 threads give +2 % fps
__global__ __launch_bounds__(SR_T) void stereo_rows_kernel(const CorbOrbParams p, const CorbStereoParams s)
{
    extern __shared__ int rows_smem[];               // cnt[rows0 + 1] | cursor[rows0]
    __shared__ int red[SR_T / 64];
    const int frame = s.frame_base + blockIdx.x, tid = threadIdx.x;
    const int R = s.rows0;
    int* cnt = rows_smem; int* cursor = rows_smem + R + 1;
    const int Nr = p.out_count[2 * frame + 1];
    const CorbKeyPoint* kr = p.out_kp + (size_t)(2 * frame + 1) * p.out_cap;
    int* row_off = s.row_off + (size_t)frame * (R + 1);
    int2* row_idx = s.row_idx + (size_t)frame * s.row_cap;
    for (int i = tid; i <= R; i += SR_T) cnt[i] = 0;
    __syncthreads();
    for (int iR = tid; iR < Nr; iR += SR_T) {
        const CorbKeyPoint k = kr[iR];
        const float r = __fmul_rn(2.0f, s.scale[k.octave]);
        const int maxr = min((int)ceilf(__fadd_rn(k.y, r)), R - 1), minr = max((int)floorf(__fsub_rn(k.y, r)), 0);
        for (int yi = minr; yi <= maxr; yi++) atomicAdd(&cnt[yi], 1);
    }
    __syncthreads();
    // exclusive scan of cnt[0..R] (serial chunks + one partial per wave)
    const int per = (R + SR_T) / SR_T;
    const int b0 = min(tid * per, R + 1), b1 = min(b0 + per, R + 1);
    int sum = 0;
    for (int i = b0; i < b1; i++) sum += cnt[i];
    int incl = sum;
    const int lane = tid & 63, wave = tid >> 6;
    incl = lx_wave_incl_scan_i(incl);
    if (lane == 63) red[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; w++) base += red[w];
    for (int i = b0; i < b1; i++) { const int t = cnt[i]; cnt[i] = base; if (i < R) cursor[i] = base; row_off[i] = base; base += t; }
    __syncthreads();
    // per left keypoint the candidate range of its row (:505-515): the matcher then needs no look-up of its own before it reads the candidates
    {
        const bool overflow = cnt[R] > s.row_cap;
        const int Nl = p.out_count[2 * frame];
        const CorbKeyPoint* kl = p.out_kp + (size_t)(2 * frame) * p.out_cap;
        int2* lr = s.left_range + (size_t)frame * p.out_cap;
        for (int iL = tid; iL < Nl; iL += SR_T) {
            const int row = (int)kl[iL].y;
            lr[iL] = (row >= 0 && row < R && !overflow) ? make_int2(cnt[row], cnt[row + 1]) : make_int2(0, 0);
        }
    }
    if (cnt[R] > s.row_cap) { if (tid == 0) p.status[2 * frame] = CORB_ERR_OVERFLOW; return; }
    for (int iR = tid; iR < Nr; iR += SR_T) {
        const CorbKeyPoint k = kr[iR];
        const float r = __fmul_rn(2.0f, s.scale[k.octave]);
        const int maxr = min((int)ceilf(__fadd_rn(k.y, r)), R - 1), minr = max((int)floorf(__fsub_rn(k.y, r)), 0);
        const int2 ent = make_int2(iR | (k.octave << 16), __float_as_int(k.x));
        for (int yi = minr; yi <= maxr; yi++) row_idx[atomicAdd(&cursor[yi], 1)] = ent;
    }
}

// Frame::ComputeStereoMatches, per left keypoint: Hamming over the row's candidates, then the 11x11 SAD sub-pixel refinement.
// SIXTEEN LANES per left keypoint, four keypoints per wavefront: the kernel is a chain of four dependent memory round trips (keypoint + range ->
// candidates -> descriptors -> image rows) with little arithmetic between them, so what counts is how many keypoints are in flight per wavefront slot
// (one keypoint per wavefront, the first form: 88 us per 128 images alone).  `alive` is per 16-lane group; the wavefront leaves when no group is.
__device__ __forceinline__ unsigned gmin16_u32(unsigned v)
{
    v = min(v, (unsigned)lx_xor_i<8>((int)v)); v = min(v, (unsigned)lx_xor_i<4>((int)v)); v = min(v, (unsigned)lx_xor_i<2>((int)v)); v = min(v, (unsigned)lx_xor_i<1>((int)v));
    return v;
}
#define SM_T 64              // threads per workgroup (the wavefronts are independent: no barrier, no LDS)
__global__ __launch_bounds__(SM_T) void stereo_match_kernel(const CorbOrbParams p, const CorbStereoParams s)
{
    int grp, frame; corb_xcd_remap(grp, frame); frame += s.frame_base;
    const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15, gbase = lane & 48;
    const int iL = grp * (SM_T / 16) + (threadIdx.x >> 6) * 4 + sub;
    const int imgL = 2 * frame, imgR = 2 * frame + 1;
    // the keypoint count, the keypoint, its descriptor and its candidate range are requested together (a slot past the count holds stale
    // but addressable data): one memory round trip before the candidates instead of three
    const int N = p.out_count[imgL];
    const int iLc = min(iL, p.out_cap - 1);
    const CorbKeyPoint kl = p.out_kp[(size_t)imgL * p.out_cap + iLc];
    const int2 crange = s.left_range[(size_t)frame * p.out_cap + iLc];
    const unsigned long long* dl = reinterpret_cast<const unsigned long long*>(p.out_desc + ((size_t)imgL * p.out_cap + iLc) * 32);
    unsigned long long a[4] = {dl[0], dl[1], dl[2], dl[3]};
    bool alive = iL < N;
    if (!__any(alive)) return;
    float* o_ur = s.u_right + (size_t)frame * p.out_cap + iLc;
    float* o_depth = s.depth + (size_t)frame * p.out_cap + iLc;
    int* o_sad = s.sad + (size_t)frame * p.out_cap + iLc;
    if (alive && sl == 0) { *o_ur = -1.0f; *o_depth = -1.0f; *o_sad = -1; }
    const int levelL = alive ? kl.octave : 0;
    const float vL = kl.y, uL = kl.x;
    const int row = (int)vL;
    alive = alive && row >= 0 && row < s.rows0;
    const float maxD = __fdiv_rn(s.bf, s.mb);                 // mbf/minZ, minZ = mb (:500-502)
    const float minU = __fsub_rn(uL, maxD), maxU = uL;        // minD = 0
    alive = alive && !(maxU < 0);
    const unsigned long long* drb = reinterpret_cast<const unsigned long long*>(p.out_desc + (size_t)imgR * p.out_cap * 32);
    const int2* row_idx = s.row_idx + (size_t)frame * s.row_cap;
    const int c0 = alive ? crange.x : 0, c1 = alive ? crange.y : 0;
    unsigned best = ((unsigned)CORB_TH_HIGH << 16) | 0xFFFFu;   // int bestDist = TH_HIGH; strict '<' => first iR wins
    float best_x = 0.f;
    // two candidates per lane and trip: both row entries, then both descriptors are requested before anything is compared
    for (int c = c0 + sl; __any(c < c1); c += 32) {
        const bool has0 = c < c1, has1 = c + 16 < c1;
        const int2 e0 = row_idx[has0 ? c : 0], e1 = row_idx[has1 ? c + 16 : 0];
        const int i0 = e0.x & 0xFFFF, i1 = e1.x & 0xFFFF, oc0 = e0.x >> 16, oc1 = e1.x >> 16;
        const float x0 = __int_as_float(e0.y), x1 = __int_as_float(e1.y);
        const bool ok0 = has0 && oc0 >= levelL - 1 && oc0 <= levelL + 1 && x0 >= minU && x0 <= maxU;
        const bool ok1 = has1 && oc1 >= levelL - 1 && oc1 <= levelL + 1 && x1 >= minU && x1 <= maxU;
        unsigned long long b0[4] = {0, 0, 0, 0}, b1[4] = {0, 0, 0, 0};
        if (ok0) { const unsigned long long* q = drb + (size_t)i0 * 4; b0[0] = q[0]; b0[1] = q[1]; b0[2] = q[2]; b0[3] = q[3]; }
        if (ok1) { const unsigned long long* q = drb + (size_t)i1 * 4; b1[0] = q[0]; b1[1] = q[1]; b1[2] = q[2]; b1[3] = q[3]; }
        if (ok0) {
            const unsigned v = ((unsigned)hamming256(a, b0) << 16) | (unsigned)i0;
            if (v < best) { best = v; best_x = x0; }
        }
        if (ok1) {
            const unsigned v = ((unsigned)hamming256(a, b1) << 16) | (unsigned)i1;
            if (v < best) { best = v; best_x = x1; }
        }
    }
    const unsigned gbest = gmin16_u32(best);
    const int bestDist = (int)(gbest >> 16);
    const int thOrbDist = (CORB_TH_HIGH + CORB_TH_LOW) / 2;
    alive = alive && bestDist < thOrbDist;
    if (!__any(alive)) return;
    // sub-pixel refinement by 11x11 SAD over incR in [-5,5] (:556-626); integer sums are exact
    const unsigned long long own = __ballot(alive && best == gbest);     // a right keypoint is listed once per row: exactly one owner per group
    const unsigned own16 = (unsigned)(own >> gbase) & 0xFFFFu;
    const float uR0 = __shfl(best_x, gbase + (own16 ? __ffs((int)own16) - 1 : 0));
    const float sf = s.inv_scale[levelL];
    const float scaleduL = roundf(__fmul_rn(kl.x, sf));
    const float scaledvL = roundf(__fmul_rn(kl.y, sf));
    const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
    int Lpitch = p.lv[0].pitch, Lw_ = p.lv[0].w, Lh_ = p.lv[0].h, Lplane = p.lv[0].plane_off;      // the level differs between the groups: select, do not index
    for (int l = 1; l < p.nlevels; l++) if (levelL == l) { Lpitch = p.lv[l].pitch; Lw_ = p.lv[l].w; Lh_ = p.lv[l].h; Lplane = p.lv[l].plane_off; }
    const uint8_t* imL = p.pyr + (size_t)imgL * p.arena_per_image + Lplane;
    const uint8_t* imR = p.pyr + (size_t)imgR * p.arena_per_image + Lplane;
    const int w = 5, Lw = 5;
    const int y0 = (int)(scaledvL - w), x0 = (int)(scaleduL - w);
    const float iniu = scaleduR0 + Lw - w;
    const float endu = scaleduR0 + Lw + w + 1;
    alive = alive && !(iniu < 0 || endu >= (float)Lw_);
    const int xr_first = (int)(scaleduR0 - Lw - w);
    // defined guard (the reference would index outside the image and throw): no match
    alive = alive && !(xr_first < 0 || x0 < 0 || y0 < 0 || y0 + 10 >= Lh_ || x0 + 10 >= Lw_);
    if (!__any(alive)) return;
    // lane sl < 11 <-> patch row sl: the 11 left bytes and the 21 right bytes (11 shifts) of the row.  |(l - cL) - (r - cR)| =
    // |(l + cR - cL + 256) - (r + 256)| on packed 16-bit halves with v_sad_u16, two pixels per instruction (integer sums are exact, so the
    // summation order is free); cL / cR = the centre pixels (row 5) of the left patch / of the right patch at that shift.
    const bool rowlane = alive && sl < 11;
    uint32_t LW[3] = {0, 0, 0}, RW[6] = {0, 0, 0, 0, 0, 0};
    if (rowlane) {
        __builtin_memcpy(LW, imL + (size_t)(y0 + sl) * Lpitch + x0, 12);
        __builtin_memcpy(RW, imR + (size_t)(y0 + sl) * Lpitch + xr_first, 24);
    }
    const uint32_t cL = ((uint32_t)__shfl((int)LW[1], gbase + 5) >> 8) & 255u;                      // left byte 5 of row 5
    const uint32_t CR[3] = {(uint32_t)__shfl((int)RW[1], gbase + 5), (uint32_t)__shfl((int)RW[2], gbase + 5), (uint32_t)__shfl((int)RW[3], gbase + 5)};   // right bytes 4 .. 15 of row 5
    uint32_t Lp[6], Rp[21];
#pragma unroll
    for (int j = 0; j < 6; j++) Lp[j] = __builtin_amdgcn_perm(0u, LW[j >> 1], 0x0c000c00u | (uint32_t)((2 * j) & 3) | ((uint32_t)(((2 * j) & 3) + 1) << 16));   // [L[2j], L[2j+1]]
    Lp[5] &= 0xFFFFu;                                           // pixel 11 does not exist
#pragma unroll
    for (int o = 0; o < 21; o++)
        Rp[o] = __builtin_amdgcn_perm(RW[min((o >> 2) + 1, 5)], RW[o >> 2], 0x0c000c00u | (uint32_t)(o & 3) | ((uint32_t)((o & 3) + 1) << 16)) + 0x01000100u;   // [R[o], R[o+1]] + 256
    int vD[16];
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const uint32_t cR = (CR[(k + 1) >> 2] >> (8 * ((k + 1) & 3))) & 255u;                       // right byte 5 + k of row 5
        const uint32_t dk = (cR + 256u - cL) * 0x10001u;
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) acc = __builtin_amdgcn_sad_u16(Lp[j] + dk, Rp[2 * j + k], acc);
        acc = __builtin_amdgcn_sad_u16(Lp[5] + (dk & 0xFFFFu), (Rp[10 + k] & 0xFFFFu), acc);
        vD[k] = rowlane ? (int)acc : 0;
    }
#pragma unroll
    for (int k = 11; k < 16; k++) vD[k] = 0;
    // transposing butterfly inside the 16-lane group: 16 values x 16 lanes -> lane sl holds the total of value sl (15 exchanges)
    int tot;
    {
        const bool h3 = sl & 8, h2 = sl & 4, h1 = sl & 2, h0 = sl & 1;
        int t8[8], t4[4], t2[2];
#pragma unroll
        for (int j = 0; j < 8; j++) t8[j] = (h3 ? vD[j + 8] : vD[j]) + lx_xor_i<8>(h3 ? vD[j] : vD[j + 8]);
#pragma unroll
        for (int j = 0; j < 4; j++) t4[j] = (h2 ? t8[j + 4] : t8[j]) + lx_xor_i<4>(h2 ? t8[j] : t8[j + 4]);
#pragma unroll
        for (int j = 0; j < 2; j++) t2[j] = (h1 ? t4[j + 2] : t4[j]) + lx_xor_i<2>(h1 ? t4[j] : t4[j + 2]);
        tot = (h0 ? t2[1] : t2[0]) + lx_xor_i<1>(h0 ? t2[0] : t2[1]);
    }
    // first minimum over the shifts (:606-611): min of (dist << 4 | shift), dist <= 121 * 766
    const unsigned kmin = gmin16_u32(sl < 11 ? (((unsigned)tot << 4) | (unsigned)sl) : 0xFFFFFFFFu);
    const int bk = (int)(kmin & 15u), bestDistS = (int)(kmin >> 4), bestinc = bk - Lw;
    const float dist1 = (float)__shfl(tot, gbase + max(bk - 1, 0)), dist2 = (float)bestDistS, dist3 = (float)__shfl(tot, gbase + min(bk + 1, 10));
    if (!alive || sl != 0) return;
    if (bestinc == -Lw || bestinc == Lw) return;
    const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
    if (deltaR < -1 || deltaR > 1) return;
    float bestuR = __fmul_rn(s.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
    float disparity = __fsub_rn(uL, bestuR);
    if (disparity >= 0 && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }
        *o_depth = __fdiv_rn(s.bf, disparity);
        *o_ur = bestuR;
        *o_sad = bestDistS;
    }
}

// median-based outlier rejection (:630-643): thDist = 1.5*1.4*median(SAD); drop SAD >= thDist.
// median = the (cnt/2)-th smallest SAD (sort + [size/2] in the reference), selected with two 256-bin histogram passes (SAD <= 121*510 < 2^16).
__device__ __forceinline__ void sf_select_digit(const int* hist, int k, int* out /* [2]: digit, k - (count below) */)
{
    // wave 0: lane l owns bins 4l .. 4l+3
    const int lane = threadIdx.x;
    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    int incl = h0 + h1 + h2 + h3;
    const int own = incl;
    incl = lx_wave_incl_scan_i(incl);
    const int below = incl - own;
    if (below <= k && k < incl) {                     // exactly one lane
        int r = k - below, d = 4 * lane;
        if (r >= h0) { r -= h0; d++; if (r >= h1) { r -= h1; d++; if (r >= h2) { r -= h2; d++; } } }
        out[0] = d; out[1] = r;
    }
}
__global__ __launch_bounds__(SR_T) void stereo_filter_kernel(const CorbOrbParams p, const CorbStereoParams s)
{
    __shared__ int hist[256];
    __shared__ int sh[4];                             // 0 matched, 1 digit, 2 rank inside the digit, 3 valid
    const int frame = s.frame_base + blockIdx.x, tid = threadIdx.x;
    const int N = p.out_count[2 * frame];
    int* sad = s.sad + (size_t)frame * p.out_cap;
    float* ur = s.u_right + (size_t)frame * p.out_cap;
    float* dp = s.depth + (size_t)frame * p.out_cap;
    if (tid < 256) hist[tid] = 0;
    if (tid < 4) sh[tid] = 0;
    __syncthreads();
    int c = 0;
    for (int i = tid; i < N; i += SR_T) { const int v = sad[i]; if (v >= 0) { c++; atomicAdd(&hist[v >> 8], 1); } }
    c = wsum_i32(c);
    if ((tid & 63) == 0 && c) atomicAdd(&sh[0], c);
    __syncthreads();
    const int cnt = sh[0];
    if (cnt == 0) { if (tid == 0) s.n_matched[frame] = 0; return; }
    if (tid < 64) sf_select_digit(hist, cnt / 2, sh + 1);
    __syncthreads();
    const int hi = sh[1], k2 = sh[2];
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += SR_T) { const int v = sad[i]; if (v >= 0 && (v >> 8) == hi) atomicAdd(&hist[v & 255], 1); }
    __syncthreads();
    if (tid < 64) sf_select_digit(hist, k2, sh + 1);
    __syncthreads();
    const float median = (float)((hi << 8) | sh[1]);
    const float thDist = __fmul_rn(__fmul_rn(1.5f, 1.4f), median);
    int valid = 0;
    for (int i = tid; i < N; i += SR_T) {
        const int v = sad[i];
        if (v >= 0) { if ((float)v < thDist) valid++; else { ur[i] = -1.0f; dp[i] = -1.0f; } }
    }
    valid = wsum_i32(valid);
    if ((tid & 63) == 0 && valid) atomicAdd(&sh[3], valid);
    __syncthreads();
    if (tid == 0) s.n_matched[frame] = sh[3];
}

void corb_launch_stereo(const CorbOrbParams& p, const CorbStereoParams& s0, int frame_base, int n_frames, hipStream_t stream, CorbProfiler* prof)
{
    CorbStereoParams s = s0; s.frame_base = frame_base;
    CORB_LAUNCH(prof, "stereo_rows_kernel", stereo_rows_kernel, dim3(n_frames), dim3(SR_T), (size_t)(2 * s.rows0 + 2) * sizeof(int), stream, p, s);
    CORB_LAUNCH(prof, "stereo_match_kernel", stereo_match_kernel, dim3((p.out_cap + SM_T / 16 - 1) / (SM_T / 16), n_frames), dim3(SM_T), 0, stream, p, s);
    CORB_LAUNCH(prof, "stereo_filter_kernel", stereo_filter_kernel, dim3(n_frames), dim3(SR_T), 0, stream, p, s);
}

// ------------------------------------------------------------------------------------------------
// Results of the frames [frame_base, frame_base + n) as ONE contiguous block per frame (CorbStereoFrameLayout, include/corb_accel.h): the low-latency form
// of Frame::Frame(stereo) (corb_stereo_frames) then needs one device-to-host transfer instead of seven per call.  Dword copies: every section is 4-byte aligned
// (a CorbKeyPoint is 7 dwords); only the valid entries of a section are moved, the rest of the block keeps what an earlier frame left there.
__global__ __launch_bounds__(256) void stereo_pack_kernel(const CorbOrbParams p, const CorbStereoParams s, uint8_t* out, CorbStereoFrameLayout lay)
{
    const int f = blockIdx.y, frame = s.frame_base + f;
    uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)f * lay.frame_bytes);
    const int nl = min(p.out_count[2 * frame], lay.capacity), nr = min(p.out_count[2 * frame + 1], lay.capacity);
    if (blockIdx.x == 0 && threadIdx.x == 0) { o[0] = (uint32_t)nl; o[1] = (uint32_t)nr; o[2] = (uint32_t)s.n_matched[frame]; o[3] = (uint32_t)(p.status[2 * frame] | p.status[2 * frame + 1]); }
    const size_t cap = (size_t)p.out_cap;
    const uint32_t* src[6] = { reinterpret_cast<const uint32_t*>(p.out_kp + (size_t)(2 * frame) * cap), reinterpret_cast<const uint32_t*>(p.out_kp + (size_t)(2 * frame + 1) * cap),
                               reinterpret_cast<const uint32_t*>(p.out_desc + (size_t)(2 * frame) * cap * 32), reinterpret_cast<const uint32_t*>(p.out_desc + (size_t)(2 * frame + 1) * cap * 32),
                               reinterpret_cast<const uint32_t*>(s.u_right + (size_t)frame * cap), reinterpret_cast<const uint32_t*>(s.depth + (size_t)frame * cap) };
    const int off[6] = { lay.off_kp_left, lay.off_kp_right, lay.off_desc_left, lay.off_desc_right, lay.off_u_right, lay.off_depth };
    const int nd[6] = { nl * 7, nr * 7, nl * 8, nr * 8, nl, nl };
    const int t = blockIdx.x * 256 + threadIdx.x, T = gridDim.x * 256;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t* dst = o + off[k] / 4;
        for (int i = t; i < nd[k]; i += T) dst[i] = src[k][i];
    }
}
void corb_launch_stereo_pack(const CorbOrbParams& p, const CorbStereoParams& s0, int frame_base, int n_frames, uint8_t* out, const CorbStereoFrameLayout& lay, hipStream_t stream, CorbProfiler* prof)
{
    CorbStereoParams s = s0; s.frame_base = frame_base;
    CORB_LAUNCH(prof, "stereo_pack_kernel", stereo_pack_kernel, dim3(16, n_frames), dim3(256), 0, stream, p, s, out, lay);
}

// ------------------------------------------------------------------------------------------------
// batched DescriptorDistance
__global__ void hamming_pairs_kernel(const unsigned long long* a, const unsigned long long* b, int n, int* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hamming256(a + (size_t)i * 4, b + (size_t)i * 4);
}
void corb_launch_hamming_pairs(const uint8_t* a, const uint8_t* b, int n, int* out, hipStream_t stream)
{
    hipLaunchKernelGGL(hamming_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned long long*>(a), reinterpret_cast<const unsigned long long*>(b), n, out);
}

// ------------------------------------------------------------------------------------------------
// SearchByBoW: one wavefront per vocabulary node common to both feature vectors.  Inside a node the
// reference's order dependence (a Frame feature can be claimed once, ORBmatcher.cc:212 / 711) is kept
// by walking the KF features serially while the 64 lanes scan the other side's features.
__device__ __forceinline__ int rot_bin(float a1, float a2)
{
    float rot = __fsub_rn(a1, a2);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, 1.0f / CORB_HISTO_LENGTH));
    if (bin == CORB_HISTO_LENGTH) bin = 0;
    return bin;
}

__global__ __launch_bounds__(256) void bow_match_kernel(CorbBowDev d)
{
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= d.n_pairs) return;
    const int na = d.pair_a[pair], nb = d.pair_b[pair];
    const int a0 = d.off1[na], a1 = d.off1[na + 1], b0 = d.off2[nb], b1 = d.off2[nb + 1];
    const int nB = b1 - b0;
    unsigned long long claimed = 0;                 // bit k: list position lane + 64k already matched
    for (int i1 = a0; i1 < a1; i1++) {
        const int idx1 = d.idx1[i1];
        if (!d.valid1[idx1]) continue;
        const unsigned long long* q = d.desc1 + (size_t)idx1 * 4;
        unsigned long long a[4] = {q[0], q[1], q[2], q[3]};
        unsigned b1key = (256u << 16) | 0xFFFFu;    // bestDist1 = 256, first position wins ties
        int b2 = 256;
        for (int k = 0, pos = lane; pos < nB; pos += 64, k++) {
            if ((claimed >> k) & 1ull) continue;
            const int idx2 = d.idx2[b0 + pos];
            if (d.variant == 1 && !d.valid2[idx2]) continue;
            const int dist = hamming256(a, d.desc2 + (size_t)idx2 * 4);
            const unsigned key = ((unsigned)dist << 16) | (unsigned)pos;
            if (key < b1key) { b2 = (int)(b1key >> 16); b1key = key; }
            else if (dist < b2) b2 = dist;
        }
        const unsigned win = wmin_u32(b1key);
        const int second = wmin_i32(b1key == win ? b2 : (int)(b1key >> 16));
        const int bestDist1 = (int)(win >> 16), bestDist2 = min(second, 256);
        const bool pass = d.variant == 0 ? (bestDist1 <= CORB_TH_LOW) : (bestDist1 < CORB_TH_LOW);
        if (pass && (float)bestDist1 < __fmul_rn(d.nnratio, (float)bestDist2)) {
            const int pos = (int)(win & 0xFFFFu);
            if ((pos & 63) == lane) claimed |= 1ull << (pos >> 6);
            if (lane == 0) {
                const int idx2 = d.idx2[b0 + pos];
                const int slot = d.variant == 0 ? idx2 : idx1;
                d.match[slot] = d.variant == 0 ? idx1 : idx2;
                if (d.check_ori) { const int bin = rot_bin(d.angle1[idx1], d.angle2[idx2]); d.bin[slot] = bin; atomicAdd(&d.hist[bin], 1); }
                atomicAdd(d.n_matches, 1);
            }
        }
    }
}

// ComputeThreeMaxima (ORBmatcher.cc:1746-1787) + removal of matches outside the 3 dominant bins
__global__ __launch_bounds__(256) void rot_finalize_kernel(int* match, int* bin, int n_slots, int* hist, int* n_matches, int stride)
{
    __shared__ int ind[3];
    if (threadIdx.x == 0) {
        int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < CORB_HISTO_LENGTH; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
            else if (s > max3) { max3 = s; i3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { i2 = -1; i3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { i3 = -1; }
        ind[0] = i1; ind[1] = i2; ind[2] = i3;
    }
    __syncthreads();
    int removed = 0;
    for (int s = threadIdx.x; s < n_slots; s += 256) {
        const int b = bin[s];
        if (b < 0 || b == ind[0] || b == ind[1] || b == ind[2]) continue;
        match[(size_t)s * stride] = -1; removed++;
    }
    if (removed) atomicSub(n_matches, removed);
}

void corb_launch_bow(const CorbBowDev& d, int n_slots, hipStream_t stream)
{
    if (d.n_pairs > 0) hipLaunchKernelGGL(bow_match_kernel, dim3((d.n_pairs + 3) / 4), dim3(256), 0, stream, d);
    if (d.check_ori) hipLaunchKernelGGL(rot_finalize_kernel, dim3(1), dim3(256), 0, stream, d.match, d.bin, n_slots, d.hist, d.n_matches, 1);
}

// ------------------------------------------------------------------------------------------------
// SearchForTriangulation: rows are independent (vbMatched2 is never set, ORBmatcher.cc:812/860), so
// one wavefront per unmatched KF1 feature; ties go to the LAST candidate (dist > bestDist rejects, :873).
__device__ __forceinline__ bool check_epipolar(const CorbKeyPoint& k1, const CorbKeyPoint& k2, const float* F, const float* sigma2_2)
{
    const float a = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, F[0]), __fmul_rn(k1.y, F[3])), F[6]);
    const float b = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, F[1]), __fmul_rn(k1.y, F[4])), F[7]);
    const float c = __fadd_rn(__fadd_rn(__fmul_rn(k1.x, F[2]), __fmul_rn(k1.y, F[5])), F[8]);
    const float num = __fadd_rn(__fadd_rn(__fmul_rn(a, k2.x), __fmul_rn(b, k2.y)), c);
    const float den = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
    if (den == 0) return false;
    const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
    return (double)dsqr < __dmul_rn(3.84, (double)sigma2_2[k2.octave]);
}

__global__ __launch_bounds__(256) void tri_match_kernel(CorbTriDev d)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= (d.n_queries_dev ? *d.n_queries_dev : d.n_queries)) return;
    const int idx1 = d.q_idx1[q], nb = d.q_node2[q];
    const bool bStereo1 = d.uright1[idx1] >= 0;
    const CorbKeyPoint k1 = d.kp1[idx1];
    const unsigned long long* qd = d.desc1 + (size_t)idx1 * 4;
    unsigned long long a[4] = {qd[0], qd[1], qd[2], qd[3]};
    const int b0 = d.off2[nb], nB = d.off2[nb + 1] - b0;
    unsigned best = 0xFFFFFFFFu;                      // (dist << 16 | 0xFFFF - pos): min dist, last position
    for (int pos = lane; pos < nB; pos += 64) {
        const int idx2 = d.idx2[b0 + pos];
        if (d.has_mp2[idx2]) continue;
        const bool bStereo2 = d.uright2[idx2] >= 0;
        if (d.only_stereo && !bStereo2) continue;
        const int dist = hamming256(a, d.desc2 + (size_t)idx2 * 4);
        if (dist > CORB_TH_LOW) continue;
        const CorbKeyPoint k2 = d.kp2[idx2];
        if (!bStereo1 && !bStereo2) {
            const float dx = __fsub_rn(d.ex, k2.x), dy = __fsub_rn(d.ey, k2.y);
            if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.0f, d.scale2[k2.octave])) continue;
        }
        if (!check_epipolar(k1, k2, d.F12, d.sigma2_2)) continue;
        best = min(best, ((unsigned)dist << 16) | (0xFFFFu - (unsigned)pos));
    }
    best = wmin_u32(best);
    if (lane == 0 && best != 0xFFFFFFFFu) {
        const int pos = (int)(0xFFFFu - (best & 0xFFFFu));
        const int idx2 = d.idx2[b0 + pos];
        d.match[idx1] = idx2;
        if (d.check_ori) { const int bin = rot_bin(k1.angle, d.kp2[idx2].angle); d.bin[idx1] = bin; atomicAdd(&d.hist[bin], 1); }
        atomicAdd(d.n_matches, 1);
    }
}

// ORBmatcher.cc:836-847: the queries are the features of KF1 in a vocabulary node both keyframes have, without a MapPoint, stereo if required
__global__ __launch_bounds__(256) void tri_queries_kernel(const int* off1, const int* idx1, const uint8_t* flags1, const float* uright1, const int* pa, const int* pb,
                                                         int n_common, int only_stereo, int* q_idx1, int* q_node2, int* counter)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_common) return;
    const int node1 = pa[k], node2 = pb[k];
    for (int i1 = off1[node1]; i1 < off1[node1 + 1]; i1++) {
        const int f = idx1[i1];
        if (flags1[f]) continue;
        if (only_stereo && !(uright1[f] >= 0)) continue;
        const int slot = atomicAdd(counter, 1);              // the order of the queries is irrelevant: every query owns match[f] of its own feature
        q_idx1[slot] = f; q_node2[slot] = node2;
    }
}
void corb_launch_tri_queries(const int* off1, const int* idx1, const uint8_t* flags1, const float* uright1, const int* pa, const int* pb, int n_common,
                             int only_stereo, int* q_idx1, int* q_node2, int* counter, hipStream_t stream)
{
    if (n_common > 0) hipLaunchKernelGGL(tri_queries_kernel, dim3((n_common + 255) / 256), dim3(256), 0, stream, off1, idx1, flags1, uright1, pa, pb, n_common, only_stereo, q_idx1, q_node2, counter);
}
void corb_launch_tri(const CorbTriDev& d, int n1, hipStream_t stream)
{
    if (d.n_queries > 0) hipLaunchKernelGGL(tri_match_kernel, dim3((d.n_queries + 3) / 4), dim3(256), 0, stream, d);
    if (d.check_ori) hipLaunchKernelGGL(rot_finalize_kernel, dim3(1), dim3(256), 0, stream, d.match, d.bin, n1, d.hist, d.n_matches, 1);
}
