// proj_kernels.hip -- projection-guided matchers of the Tracking thread on gfx950:
//   ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)      (C/src/ORBmatcher.cc:45-131)
//   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)           (C/src/ORBmatcher.cc:1470-1614)
//   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea               (C/src/Frame.cc:230-245, 331-395)
//
// Both routines are greedy and ORDER DEPENDENT in the reference: query i (a map point / a feature of the last frame)
// may only take a feature that no earlier query has claimed.  Here the expensive part is data parallel -- one wavefront
// per query gathers the grid cells of its search window and computes all Hamming distances once -- and the order
// dependence is resolved exactly by rounds inside one workgroup: in a round a query is FINAL when it is the lowest-indexed
// unfinished query touching every one of its still-free candidates (then no earlier query can change its outcome); final
// queries pick best / second best among their free candidates and claim.  The lowest unfinished query is always final, so
// the loop terminates; spatially scattered queries finish in a handful of rounds.
#include "proj_internal.h"
#include "lane_exchange.h"

__device__ __forceinline__ unsigned long long wmin_u64(unsigned long long v)
{
    return lx_wave_min_u64(v);
}
__device__ __forceinline__ int hamming256p(const unsigned long long* a, const unsigned long long* b)
{
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// Frame::AssignFeaturesToGrid: CSR over the 64 x 48 cells (one workgroup; order inside a cell is irrelevant because the
// matchers order candidates by (ix, iy, feature index) keys)
__global__ __launch_bounds__(1024) void proj_grid_kernel(CorbProjDev d)
{
    __shared__ int cnt[PROJ_CELLS + 1];
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    for (int c = tid; c <= PROJ_CELLS; c += 1024) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < d.n; i += 1024) {
        const CorbKeyPoint k = d.keys[i];
        const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, d.min_x), d.winv));        // PosInGrid
        const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, d.min_y), d.hinv));
        const int cell = (px < 0 || px >= PROJ_COLS || py < 0 || py >= PROJ_ROWS) ? -1 : px * PROJ_ROWS + py;
        d.feat_cell[i] = cell;
        if (cell >= 0) atomicAdd(&cnt[cell], 1);
    }
    __syncthreads();
    // exclusive scan of 3072 counts: 3 per thread + block scan
    int loc[3], s = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) { loc[j] = cnt[tid * 3 + j]; s += loc[j]; }
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { const int t = tid >= o ? part[tid - o] : 0; __syncthreads(); part[tid] += t; __syncthreads(); }
    int base = part[tid] - s;
#pragma unroll
    for (int j = 0; j < 3; j++) { d.cell_off[tid * 3 + j] = base; cnt[tid * 3 + j] = base; base += loc[j]; }
    if (tid == 1023) d.cell_off[PROJ_CELLS] = base;
    __syncthreads();
    for (int i = tid; i < d.n; i += 1024) { const int cell = d.feat_cell[i]; if (cell >= 0) d.cell_idx[atomicAdd(&cnt[cell], 1)] = i; }
}

// query preparation, SearchByProjection(Frame, MapPoints): window from RadiusByViewingCos and the predicted level
__global__ __launch_bounds__(256) void proj_prepare_map_kernel(CorbProjDev d, const CorbTrackedPoint* mp, float th)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= d.nq) return;
    const CorbTrackedPoint p = mp[q];
    CorbProjQuery o;
    float r = ((double)p.view_cos > 0.998) ? 2.5f : 4.0f;
    if (th != 1.0f) r = __fmul_rn(r, th);
    o.x = p.proj_x; o.y = p.proj_y; o.r = __fmul_rn(r, d.scale[p.level]);
    o.min_level = p.level - 1; o.max_level = p.level; o.ur_ref = p.proj_xr;
    o.valid = p.valid; o.claims = p.claims; o.angle = 0.f;
    d.query[q] = o;
}

// query preparation, SearchByProjection(Frame, Frame): project the last frame's map points into the current frame
__global__ __launch_bounds__(256) void proj_prepare_frame_kernel(CorbProjDev d, const CorbLastPoint* last, CorbProjPose pose, float th)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= d.nq) return;
    const CorbLastPoint p = last[q];
    CorbProjQuery o; o.valid = 0; o.claims = p.claims; o.angle = p.angle; o.x = o.y = o.r = o.ur_ref = 0.f; o.min_level = o.max_level = 0;
    if (p.valid) {
        float x3[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {        // x3Dc = Rcw*x3Dw + tcw : cv::gemm on CV_32F = double accumulation, one rounding
            const double s = __fma_rn((double)pose.Tcw[i * 4 + 2], (double)p.world[2], __fma_rn((double)pose.Tcw[i * 4 + 1], (double)p.world[1], __dmul_rn((double)pose.Tcw[i * 4], (double)p.world[0])));
            x3[i] = (float)__dadd_rn(s, (double)pose.Tcw[i * 4 + 3]);
        }
        const float invzc = (float)(1.0 / (double)x3[2]);
        if (!(invzc < 0)) {
            const float u = __fadd_rn(__fmul_rn(__fmul_rn(pose.fx, x3[0]), invzc), pose.cx);
            const float v = __fadd_rn(__fmul_rn(__fmul_rn(pose.fy, x3[1]), invzc), pose.cy);
            if (!(u < d.min_x || u > d.max_x || v < d.min_y || v > d.max_y)) {
                o.valid = 1; o.x = u; o.y = v; o.r = __fmul_rn(th, d.scale[p.octave]);
                o.ur_ref = __fsub_rn(u, __fmul_rn(pose.bf, invzc));
                if (pose.forward) { o.min_level = p.octave; o.max_level = -1; }
                else if (pose.backward) { o.min_level = 0; o.max_level = p.octave; }
                else { o.min_level = p.octave - 1; o.max_level = p.octave + 1; }
            }
        }
    }
    d.query[q] = o;
}

// GetFeaturesInArea + DescriptorDistance for every candidate of a query (one wavefront per query).  Lanes sweep the cells of
// the window; a candidate is stored as key = dist << 40 | ix << 32 | iy << 24 | feature (the reference's visiting order) and its octave.
__global__ __launch_bounds__(256) void proj_candidates_kernel(CorbProjDev d)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= d.nq) return;
    const CorbProjQuery Q = d.query[q];
    const int cap = d.cand_cap ? d.cand_cap : PROJ_CAND_CAP;
    int total = 0;
    if (Q.valid) {
        int x0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.x, d.min_x), Q.r), d.winv)); x0 = max(x0, 0);
        int x1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.x, d.min_x), Q.r), d.winv)); x1 = min(x1, PROJ_COLS - 1);
        int y0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.y, d.min_y), Q.r), d.hinv)); y0 = max(y0, 0);
        int y1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.y, d.min_y), Q.r), d.hinv)); y1 = min(y1, PROJ_ROWS - 1);
        if (x0 < PROJ_COLS && x1 >= 0 && y0 < PROJ_ROWS && y1 >= 0) {
            const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
            const unsigned long long* qd = d.qdesc + (size_t)q * 4;
            const unsigned long long a[4] = {qd[0], qd[1], qd[2], qd[3]};
            const int ny = y1 - y0 + 1, ncell = (x1 - x0 + 1) * ny;
            for (int c0 = 0; c0 < ncell; c0 += 64) {
                const int c = c0 + lane;
                int beg = 0, end = 0, ix = 0, iy = 0;
                if (c < ncell) { ix = x0 + c / ny; iy = y0 + c % ny; beg = d.cell_off[ix * PROJ_ROWS + iy]; end = d.cell_off[ix * PROJ_ROWS + iy + 1]; }
                int more = end - beg;
                for (int j = 0; __any(j < more); j++) {
                    bool ok = false; unsigned long long key = 0; int oct = 0;
                    if (j < more) {
                        const int f = d.cell_idx[beg + j];
                        const CorbKeyPoint k = d.keys[f];
                        oct = k.octave;
                        ok = true;
                        if (check_levels) { if (oct < Q.min_level) ok = false; if (Q.max_level >= 0 && oct > Q.max_level) ok = false; }
                        if (ok) ok = fabsf(__fsub_rn(k.x, Q.x)) < Q.r && fabsf(__fsub_rn(k.y, Q.y)) < Q.r;
                        if (ok && d.check_uright) { const float ur = d.u_right[f]; if (ur > 0 && fabsf(__fsub_rn(Q.ur_ref, ur)) > Q.r) ok = false; }
                        if (ok) {
                            const int dist = hamming256p(a, d.desc + (size_t)f * 4);
                            key = ((unsigned long long)dist << 40) | ((unsigned long long)ix << 32) | ((unsigned long long)iy << 24) | (unsigned long long)f;
                        }
                    }
                    const unsigned long long m = __ballot(ok);
                    if (ok) {
                        const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
                        if (pos < cap) { d.cand_key[(size_t)q * cap + pos] = key; d.cand_oct[(size_t)q * cap + pos] = (unsigned char)oct; }
                    }
                    total += __popcll(m);
                }
            }
        }
    }
    if (lane == 0) { d.cand_cnt[q] = min(total, cap); if (total > cap) *d.status = CORB_ERR_OVERFLOW; }
}

// exact resolution of the greedy, order-dependent assignment (one workgroup, thread per query, rounds)
__global__ __launch_bounds__(1024) void proj_resolve_kernel(CorbProjDev d)
{
    extern __shared__ int lds[];
    int* feat_min = lds;                                  // [n]   lowest unfinished query touching the feature
    int* match = feat_min + d.n;                          // [n]   assigned query (atomicMax), -1 = none
    unsigned char* claimed = reinterpret_cast<unsigned char*>(match + d.n);      // [n]
    unsigned char* fin = claimed + ((d.n + 3) & ~3);      // [nq]
    __shared__ int remaining, nmatches, hist[CORB_HISTO_LENGTH], ind[3];
    const int tid = threadIdx.x;
    for (int f = tid; f < d.n; f += 1024) { match[f] = -1; claimed[f] = d.claimed[f]; }
    for (int q = tid; q < d.nq; q += 1024) { fin[q] = (!d.query[q].valid || d.cand_cnt[q] == 0) ? 1 : 0; d.ev_feat[q] = -1; }
    if (tid < CORB_HISTO_LENGTH) hist[tid] = 0;
    if (tid == 0) nmatches = 0;
    int my_matches = 0;
    __syncthreads();
    // A thread's first two queries (q = tid, tid + 1024: all of them up to 2 048 queries) keep their first PR_RC candidates in registers for the whole call -- the rounds
    // below used to re-read every unfinished query's list from global memory twice per round, one dependent load per candidate (95 us of a tracked frame's
    // SearchByProjection at 1 750 queries); the rest of a long list, and further queries, still come from memory.
#define PR_RC 12
    unsigned long long rk[2][PR_RC]; unsigned long long ro[2] = {0ull, 0ull}; int rn[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int q = tid + 1024 * j;
        const int nc = q < d.nq ? d.cand_cnt[q] : 0;
        rn[j] = nc;
        const unsigned long long* ck = d.cand_key + (size_t)(q < d.nq ? q : 0) * PROJ_CAND_CAP;
        const unsigned char* co = d.cand_oct + (size_t)(q < d.nq ? q : 0) * PROJ_CAND_CAP;
#pragma unroll
        for (int c = 0; c < PR_RC; c++) { rk[j][c] = c < nc ? ck[c] : ~0ull; if (c < nc) ro[j] |= (unsigned long long)(co[c] & 15) << (4 * c); }
    }
    for (;;) {
        if (tid == 0) remaining = 0;
        for (int f = tid; f < d.n; f += 1024) feat_min[f] = 0x7FFFFFFF;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int q = tid + 1024 * j;
            if (q >= d.nq || fin[q]) continue;
#pragma unroll
            for (int c = 0; c < PR_RC; c++) if (c < rn[j]) { const int f = (int)(rk[j][c] & 0xFFFFFFull); if (!claimed[f]) atomicMin(&feat_min[f], q); }
            const unsigned long long* ck = d.cand_key + (size_t)q * PROJ_CAND_CAP;
            for (int c = PR_RC; c < rn[j]; c++) { const int f = (int)(ck[c] & 0xFFFFFFull); if (!claimed[f]) atomicMin(&feat_min[f], q); }
        }
        for (int q = tid + 2048; q < d.nq; q += 1024) {
            if (fin[q]) continue;
            const unsigned long long* ck = d.cand_key + (size_t)q * PROJ_CAND_CAP;
            const int nc = d.cand_cnt[q];
            for (int c = 0; c < nc; c++) { const int f = (int)(ck[c] & 0xFFFFFFull); if (!claimed[f]) atomicMin(&feat_min[f], q); }
        }
        __syncthreads();
        for (int jq = 0, q = tid; q < d.nq; q += 1024, jq++) {
            if (fin[q]) continue;
            const unsigned long long* ck = d.cand_key + (size_t)q * PROJ_CAND_CAP;
            const unsigned char* co = d.cand_oct + (size_t)q * PROJ_CAND_CAP;
            unsigned long long k1 = ~0ull, k2 = ~0ull; int o1 = -1, o2 = -1;
            int c0 = 0, nc;
            if (jq < 2) {                                                // the cached part of the list (same order, same comparisons)
                nc = jq == 0 ? rn[0] : rn[1];
                const unsigned long long oo = jq == 0 ? ro[0] : ro[1];
#pragma unroll
                for (int c = 0; c < PR_RC; c++) {
                    const unsigned long long k = jq == 0 ? rk[0][c] : rk[1][c];
                    if (c < nc) {
                        const int f = (int)(k & 0xFFFFFFull);
                        if (!claimed[f]) { const int oc = (int)((oo >> (4 * c)) & 15); if (k < k1) { k2 = k1; o2 = o1; k1 = k; o1 = oc; } else if (k < k2) { k2 = k; o2 = oc; } }
                    }
                }
                c0 = PR_RC;
            } else nc = d.cand_cnt[q];
            for (int c = c0; c < nc; c++) {
                const unsigned long long k = ck[c];
                const int f = (int)(k & 0xFFFFFFull);
                if (claimed[f]) continue;
                if (k < k1) { k2 = k1; o2 = o1; k1 = k; o1 = co[c]; } else if (k < k2) { k2 = k; o2 = co[c]; }
            }
            // the decision is a function of the best two unclaimed candidates only: it is final once no earlier unfinished query can still claim
            // either of them (claims by earlier queries on the other candidates leave the best two what they are; later queries never claim a
            // candidate of an earlier unfinished one).  Waiting for ALL candidates to be free of earlier queries took 3-4x the rounds.
            // (without the ratio test -- SearchByProjection(Frame, Frame) -- the second best plays no part: only the best candidate has to be free of earlier queries;
            // at 2 000 queries this halves the rounds)
            const bool is_final = (k1 == ~0ull || feat_min[(int)(k1 & 0xFFFFFFull)] == q) && (!d.ratio_test || k2 == ~0ull || feat_min[(int)(k2 & 0xFFFFFFull)] == q);
            if (!is_final) { remaining = 1; continue; }                  // (a flag: up to 2 000 atomic adds on the one LDS word per round were most of a round's time)
            fin[q] = 1;
            if (k1 == ~0ull) continue;                                   // every candidate is taken
            const int bestDist = (int)(k1 >> 40), bestDist2 = k2 == ~0ull ? 256 : (int)(k2 >> 40);
            if (bestDist > d.th_dist) continue;
            if (d.ratio_test && o1 == o2 && (float)bestDist > __fmul_rn(d.nnratio, (float)bestDist2)) continue;   // bestLevel==bestLevel2 (-1 == -1 never: o1 >= 0)
            const int f = (int)(k1 & 0xFFFFFFull);
            atomicMax(&match[f], q);
            if (d.query[q].claims) claimed[f] = 1;                       // visible to later rounds (no other final query of this round touches f)
            my_matches++;
            if (d.check_ori) {
                float rot = __fsub_rn(d.query[q].angle, d.keys[f].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / CORB_HISTO_LENGTH));
                if (bin == CORB_HISTO_LENGTH) bin = 0;
                d.ev_feat[q] = f; d.ev_bin[q] = bin; atomicAdd(&hist[bin], 1);
            }
        }
        __syncthreads();
        if (remaining == 0) break;
        __syncthreads();
    }
    {   // the thread's matches: one add per wavefront
        int mm = my_matches;
        mm = lx_wave_sum_i(mm);
        if ((tid & 63) == 0 && mm) atomicAdd(&nmatches, mm);
        __syncthreads();
    }
#undef PR_RC
    if (d.check_ori) {
        if (tid == 0) {                                                  // ComputeThreeMaxima (ORBmatcher.cc:1746-1787)
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
        for (int q = tid; q < d.nq; q += 1024) {
            const int f = d.ev_feat[q];
            if (f < 0) continue;
            const int b = d.ev_bin[q];
            if (b != ind[0] && b != ind[1] && b != ind[2]) { match[f] = -1; atomicSub(&nmatches, 1); }
        }
        __syncthreads();
    }
    for (int f = tid; f < d.n; f += 1024) d.match[f] = match[f];
    if (tid == 0) *d.n_matches = nmatches;
}

// ------------------------------------------------------------------------------------------------------------------
// Keyframe-target matchers: SearchByProjection(Frame&, KeyFrame*, ...) (ORBmatcher.cc:1616-1744), Fuse x2 (:960-1241),
// SearchBySim3 (:1244-1468).  Query = a MapPoint; its projection, the distance / viewing-angle gates, PredictScale and
// the search radius are computed here with the reference's float arithmetic (3x3 products = cv::gemm: double
// accumulation, one rounding; cv::norm / Mat::dot = double sums).  PredictScale's libm log(float) is DEFINED as
// (float)log((double)ratio).
__device__ __forceinline__ void proj_gemm3(const float* M /* 3x4 */, const float* x, float* o)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double s = __fma_rn((double)M[i * 4 + 2], (double)x[2], __fma_rn((double)M[i * 4 + 1], (double)x[1], __dmul_rn((double)M[i * 4], (double)x[0])));
        o[i] = (float)__dadd_rn(s, (double)M[i * 4 + 3]);
    }
}
__device__ __forceinline__ float proj_norm3(const float* v)
{
    return (float)sqrt(__fma_rn((double)v[2], (double)v[2], __fma_rn((double)v[1], (double)v[1], __dmul_rn((double)v[0], (double)v[0]))));
}

__global__ __launch_bounds__(256) void proj_prepare_points_kernel(CorbProjDev d, const CorbMapPointView* pts, CorbProjTf tf)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= d.nq) return;
    const CorbMapPointView p = pts[q];
    CorbProjQuery o; o.valid = 0; o.claims = 1; o.angle = p.angle; o.x = o.y = o.r = o.ur_ref = 0.f; o.min_level = o.max_level = 0;
    if (p.valid) {
        float pa[3], pc[3];
        proj_gemm3(tf.A, p.world, pa);
        if (tf.two) proj_gemm3(tf.B, pa, pc); else { pc[0] = pa[0]; pc[1] = pa[1]; pc[2] = pa[2]; }
        bool ok = tf.reloc ? true : !(pc[2] < 0.0f);
        const float invz = tf.invz_double ? (float)(1.0 / (double)pc[2]) : __fdiv_rn(1.0f, pc[2]);
        float u, v;
        if (tf.reloc) { u = __fadd_rn(__fmul_rn(__fmul_rn(tf.fx, pc[0]), invz), tf.cx); v = __fadd_rn(__fmul_rn(__fmul_rn(tf.fy, pc[1]), invz), tf.cy); }
        else { u = __fadd_rn(__fmul_rn(tf.fx, __fmul_rn(pc[0], invz)), tf.cx); v = __fadd_rn(__fmul_rn(tf.fy, __fmul_rn(pc[1], invz)), tf.cy); }
        if (tf.reloc) { if (u < d.min_x || u > d.max_x || v < d.min_y || v > d.max_y) ok = false; }
        else if (!(u >= d.min_x && u < d.max_x && v >= d.min_y && v < d.max_y)) ok = false;              // KeyFrame::IsInImage
        if (ok) {
            const float PO[3] = { __fsub_rn(p.world[0], tf.Ow[0]), __fsub_rn(p.world[1], tf.Ow[1]), __fsub_rn(p.world[2], tf.Ow[2]) };
            const float dist3D = tf.dist_from_cam ? proj_norm3(pc) : proj_norm3(PO);
            const float maxD = __fmul_rn(1.2f, p.max_distance), minD = __fmul_rn(0.8f, p.min_distance);
            if (dist3D < minD || dist3D > maxD) ok = false;
            if (ok && tf.check_normal) {
                const double dot = __fma_rn((double)PO[2], (double)p.normal[2], __fma_rn((double)PO[1], (double)p.normal[1], __dmul_rn((double)PO[0], (double)p.normal[0])));
                if (dot < __dmul_rn(0.5, (double)dist3D)) ok = false;
            }
            if (ok) {
                const float ratio = __fdiv_rn(p.max_distance, dist3D);
                const float lg = (float)log((double)ratio);
                int lvl = (int)ceilf(__fdiv_rn(lg, tf.log_scale));
                if (lvl < 0) lvl = 0; else if (lvl >= tf.nlevels) lvl = tf.nlevels - 1;
                o.valid = 1; o.x = u; o.y = v; o.r = __fmul_rn(tf.th, d.scale[lvl]);
                o.ur_ref = __fsub_rn(u, __fmul_rn(tf.bf, invz));
                o.min_level = lvl - 1; o.max_level = lvl + tf.lvl_hi;
            }
        }
    }
    d.query[q] = o;
}

// independent best candidate of every query (one wavefront per query): KeyFrame::GetFeaturesInArea(u, v, r) + the octave
// window, optionally the reprojection chi2 test of Fuse (:1040-1066), first-minimum in the reference's visiting order
__global__ __launch_bounds__(256) void proj_best_kernel(CorbProjDev d)
{
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (q >= d.nq) return;
    const CorbProjQuery Q = d.query[q];
    unsigned long long best = ~0ull;
    if (Q.valid) {
        int x0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.x, d.min_x), Q.r), d.winv)); x0 = max(x0, 0);
        int x1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.x, d.min_x), Q.r), d.winv)); x1 = min(x1, PROJ_COLS - 1);
        int y0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(Q.y, d.min_y), Q.r), d.hinv)); y0 = max(y0, 0);
        int y1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(Q.y, d.min_y), Q.r), d.hinv)); y1 = min(y1, PROJ_ROWS - 1);
        if (x0 < PROJ_COLS && x1 >= 0 && y0 < PROJ_ROWS && y1 >= 0) {
            const unsigned long long* qd = d.qdesc + (size_t)q * 4;
            const unsigned long long a[4] = {qd[0], qd[1], qd[2], qd[3]};
            const int ny = y1 - y0 + 1, ncell = (x1 - x0 + 1) * ny;
            for (int c0 = 0; c0 < ncell; c0 += 64) {
                const int c = c0 + lane;
                if (c >= ncell) continue;
                const int ix = x0 + c / ny, iy = y0 + c % ny;
                const int beg = d.cell_off[ix * PROJ_ROWS + iy], end = d.cell_off[ix * PROJ_ROWS + iy + 1];
                for (int j = beg; j < end; j++) {
                    const int f = d.cell_idx[j];
                    const CorbKeyPoint k = d.keys[f];
                    if (!(fabsf(__fsub_rn(k.x, Q.x)) < Q.r && fabsf(__fsub_rn(k.y, Q.y)) < Q.r)) continue;
                    if (k.octave < Q.min_level || k.octave > Q.max_level) continue;
                    if (d.chi2_check) {
                        const float ex = __fsub_rn(Q.x, k.x), ey = __fsub_rn(Q.y, k.y);
                        const float ur = d.u_right[f];
                        if (ur >= 0) {
                            const float er = __fsub_rn(Q.ur_ref, ur);
                            const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                            if ((double)__fmul_rn(e2, d.inv_sigma2[k.octave]) > 7.8) continue;
                        } else {
                            const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                            if ((double)__fmul_rn(e2, d.inv_sigma2[k.octave]) > 5.99) continue;
                        }
                    }
                    const int dist = hamming256p(a, d.desc + (size_t)f * 4);
                    const unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)ix << 32) | ((unsigned long long)iy << 24) | (unsigned long long)f;
                    best = key < best ? key : best;
                }
            }
        }
    }
    best = wmin_u64(best);
    if (lane == 0) {
        const int dist = best == ~0ull ? 256 : (int)(best >> 40);
        d.best_dist[q] = dist;
        d.best_idx[q] = (best != ~0ull && dist <= d.th_dist) ? (int)(best & 0xFFFFFFull) : -1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:540-655): the monocular initialiser's matcher.
// Queries = the level-0 features of F1, window = windowSize around vbPrevMatched[i1], candidates = level-0 features of F2 (GetFeaturesInArea(x, y, windowSize, 0, 0)):
// proj_grid_kernel + proj_candidates_kernel above produce every query's candidate list in the reference's visiting order.  The loop over i1 itself is sequential by
// construction -- a candidate is skipped while its current partner is at least as close (:583), so which candidates a feature sees depends on every commit before it, and
// a commit can take a feature away from an earlier one (:601-605) -- and it runs as ONE wavefront: per query the lanes test the candidates against the LDS table of
// current match distances, a wavefront minimum gives the best key (distance, then visiting order) and the second-best distance, lane 0 commits.  ~2 000 queries x a few
// hundred cycles: the call is a one-off at start-up (Tracking::MonocularInitialization), exactness is what matters.
__global__ __launch_bounds__(256) void init_prepare_kernel(CorbProjDev d, const CorbKeyPoint* keys1, const float* prev_matched, float window)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= d.nq) return;
    CorbProjQuery o; o.valid = keys1[q].octave > 0 ? 0 : 1; o.claims = 1; o.angle = keys1[q].angle;      // if(level1>0) continue; (:558-560)
    o.x = prev_matched[2 * q]; o.y = prev_matched[2 * q + 1]; o.r = window; o.ur_ref = 0.f; o.min_level = 0; o.max_level = 0;
    d.query[q] = o;
}
__global__ __launch_bounds__(64) void init_resolve_kernel(CorbProjDev d, const CorbKeyPoint* keys1, float* prev_matched)
{
    extern __shared__ int lds[];
    int* matched_dist = lds;                              // [n]  vMatchedDistance
    int* match21 = lds + d.n;                             // [n]  vnMatches21
    __shared__ int hist[CORB_HISTO_LENGTH], ind[3], nmatches;
    const int lane = threadIdx.x;
    for (int f = lane; f < d.n; f += 64) { matched_dist[f] = 0x7FFFFFFF; match21[f] = -1; }
    if (lane < CORB_HISTO_LENGTH) hist[lane] = 0;
    if (lane == 0) nmatches = 0;
    for (int q = lane; q < d.nq; q += 64) { d.best_idx[q] = -1; d.ev_bin[q] = -1; }          // best_idx = vnMatches12
    __syncthreads();
    for (int q = 0; q < d.nq; q++) {
        const int cnt = d.query[q].valid ? d.cand_cnt[q] : 0;                                  // (the same address on every lane)
        if (cnt == 0) continue;
        unsigned long long lbest = ~0ull; int lsecond = 0x7FFFFFFF;
        for (int j = lane; j < cnt; j += 64) {
            const unsigned long long key = d.cand_key[(size_t)q * d.cand_cap + j];
            const int dist = (int)(key >> 40), f = (int)(key & 0xFFFFFFull);
            if (matched_dist[f] <= dist) continue;                                             // if(vMatchedDistance[i2]<=dist) continue; (:583)
            if (key < lbest) { if (lbest != ~0ull) lsecond = min(lsecond, (int)(lbest >> 40)); lbest = key; }
            else lsecond = min(lsecond, dist);
        }
        const unsigned long long best = wmin_u64(lbest);
        // bestDist2 = the second smallest distance of the remaining candidates (a multiset: :586-595): the winner's lane offers its own second, every other lane its best
        int second = (lbest == best) ? lsecond : (int)(lbest >> 40);
        if (lbest == ~0ull) second = 0x7FFFFFFF;
        second = lx_wave_min_i(second);
        if (lane == 0 && best != ~0ull) {
            const int bestDist = (int)(best >> 40), bestIdx2 = (int)(best & 0xFFFFFFull);
            if (bestDist <= CORB_TH_LOW && (float)bestDist < __fmul_rn((float)second, d.nnratio)) {        // (:597-599)
                if (match21[bestIdx2] >= 0) { d.best_idx[match21[bestIdx2]] = -1; nmatches--; }
                d.best_idx[q] = bestIdx2; match21[bestIdx2] = q; matched_dist[bestIdx2] = bestDist; nmatches++;
                if (d.check_ori) {
                    float rot = __fsub_rn(keys1[q].angle, d.keys[bestIdx2].angle);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    int bin = (int)roundf(__fmul_rn(rot, 1.0f / CORB_HISTO_LENGTH));
                    if (bin == CORB_HISTO_LENGTH) bin = 0;
                    d.ev_bin[q] = bin; hist[bin]++;                                            // (a partner lost later stays in the histogram: rotHist keeps its entry)
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __syncthreads();
    if (d.check_ori) {
        if (lane == 0) {                                                                       // ComputeThreeMaxima (ORBmatcher.cc:1746-1787)
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
        int lost = 0;
        for (int q = lane; q < d.nq; q += 64) {
            const int b = d.ev_bin[q];
            if (b >= 0 && b != ind[0] && b != ind[1] && b != ind[2] && d.best_idx[q] >= 0) { d.best_idx[q] = -1; lost++; }      // (:636-646)
        }
        lost = lx_wave_sum_i(lost);
        if (lane == 0) nmatches -= lost;
        __syncthreads();
    }
    for (int q = lane; q < d.nq; q += 64) {                                                    // Update prev matched (:650-653)
        const int m = d.best_idx[q];
        if (m >= 0) { prev_matched[2 * q] = d.keys[m].x; prev_matched[2 * q + 1] = d.keys[m].y; }
    }
    if (lane == 0) *d.n_matches = nmatches;
}
void corb_launch_search_for_initialization(const CorbProjDev& d, const CorbKeyPoint* keys1, float* prev_matched, float window, hipStream_t s)
{
    hipLaunchKernelGGL(proj_grid_kernel, dim3(1), dim3(1024), 0, s, d);
    if (d.nq > 0) {
        hipLaunchKernelGGL(init_prepare_kernel, dim3((d.nq + 255) / 256), dim3(256), 0, s, d, keys1, (const float*)prev_matched, window);
        hipLaunchKernelGGL(proj_candidates_kernel, dim3((d.nq + 3) / 4), dim3(256), 0, s, d);
    }
    hipLaunchKernelGGL(init_resolve_kernel, dim3(1), dim3(64), (size_t)d.n * 8 + 16, s, d, keys1, prev_matched);
}

void corb_launch_projection_points(const CorbProjDev& d, const CorbMapPointView* pts, const CorbProjTf& tf, int greedy, hipStream_t s)
{
    hipLaunchKernelGGL(proj_grid_kernel, dim3(1), dim3(1024), 0, s, d);
    if (d.nq > 0) hipLaunchKernelGGL(proj_prepare_points_kernel, dim3((d.nq + 255) / 256), dim3(256), 0, s, d, pts, tf);
    if (greedy) {
        if (d.nq > 0) hipLaunchKernelGGL(proj_candidates_kernel, dim3((d.nq + 3) / 4), dim3(256), 0, s, d);
        const size_t lds = (size_t)d.n * 8 + ((d.n + 3) & ~3) + ((d.nq + 3) & ~3) + 16;
        hipLaunchKernelGGL(proj_resolve_kernel, dim3(1), dim3(1024), lds, s, d);
    } else if (d.nq > 0)
        hipLaunchKernelGGL(proj_best_kernel, dim3((d.nq + 3) / 4), dim3(256), 0, s, d);
}

void corb_launch_projection(const CorbProjDev& d, const CorbTrackedPoint* mp, const CorbLastPoint* last, const CorbProjPose* pose, float th, hipStream_t s)
{
    hipLaunchKernelGGL(proj_grid_kernel, dim3(1), dim3(1024), 0, s, d);
    if (d.nq > 0) {
        if (mp) hipLaunchKernelGGL(proj_prepare_map_kernel, dim3((d.nq + 255) / 256), dim3(256), 0, s, d, mp, th);
        else hipLaunchKernelGGL(proj_prepare_frame_kernel, dim3((d.nq + 255) / 256), dim3(256), 0, s, d, last, *pose, th);
        hipLaunchKernelGGL(proj_candidates_kernel, dim3((d.nq + 3) / 4), dim3(256), 0, s, d);
    }
    const size_t lds = (size_t)d.n * 8 + ((d.n + 3) & ~3) + ((d.nq + 3) & ~3) + 16;
    hipLaunchKernelGGL(proj_resolve_kernel, dim3(1), dim3(1024), lds, s, d);
}
