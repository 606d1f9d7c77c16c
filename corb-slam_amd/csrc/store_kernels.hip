// store_kernels.hip -- the bundle-adjustment graph of Optimizer::BundleAdjustment (C/src/Optimizer.cc:54-270) built on the device from store records:
// vertices from the keyframe / map-point headers, edges from the map points' observation lists (mObservations: keyframe id -> feature index) resolved
// through a device id table, measurements from the observing keyframe's record -- and the write-back of the estimates into the records (:216-262).
#include "store_internal.h"
#include "lane_exchange.h"
#include "device_util.h"
#include "ba_store_internal.h"

// per keyframe of the problem: pose, intrinsics, fixed / bad flags; id -> vertex index into the table
__global__ __launch_bounds__(256) void bas_kf_kernel(BAStoreDev d)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.n_kf) return;
    const KfHeader* h = reinterpret_cast<const KfHeader*>(d.kf_base + (size_t)d.kf_slots[i] * d.kf_bytes);
    float* T = d.poses + 16 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 16; k++) T[k] = h->m.Tcw[k];
    float* c = d.intr + 5 * (size_t)i;
    c[0] = h->m.fx; c[1] = h->m.fy; c[2] = h->m.cx; c[3] = h->m.cy; c[4] = h->m.bf;
    const bool bad = (h->m.flags & CORB_KF_BAD) != 0;
    d.kf_bad[i] = bad ? 1 : 0;
    // vSE3->setFixed(pKF->mnId==1 || pKF->getFixed()) (Optimizer.cc:92); a bad keyframe is no vertex (:86-87): it has no edges here and is passed through
    d.pose_fixed[i] = (bad || h->m.id == 1ull || (h->m.flags & CORB_KF_FIXED) || i >= d.n_local) ? 1 : 0;       // (lFixedCameras: vSE3->setFixed(true), :582)
    if (!corb_idtab_insert(d.tab, h->m.id, i)) atomicOr(d.status, BAS_DUPLICATE_KF);
}

// edges of one map point (count pass / fill pass): mObservations in record order (ascending keyframe id, MapPoint.h:182)
template <bool FILL>
__global__ __launch_bounds__(256) void bas_mp_kernel(BAStoreDev d)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d.n_mp) return;
    const char* rec = d.mp_base + (size_t)d.mp_slots[j] * d.mp_bytes;
    const CorbMapPointRecord* h = reinterpret_cast<const CorbMapPointRecord*>(rec);
    const bool bad = (h->flags & CORB_MP_BAD) != 0;
    if (!FILL) {
        float* p = d.points + 3 * (size_t)j;
        p[0] = h->world_pos[0]; p[1] = h->world_pos[1]; p[2] = h->world_pos[2];
        d.point_fixed[j] = (h->flags & CORB_MP_FIXED) ? 1 : 0;            // vPoint->setFixed(pMP->getFixed()) (:120)
        d.mp_bad[j] = bad ? 1 : 0;
    }
    int cnt = 0;
    if (!bad) {                                                            // if(pMP->isBad()) continue; (:108-109)
        const MpLayout L(d.max_obs);
        const unsigned long long* okf = reinterpret_cast<const unsigned long long*>(rec + L.obs_kf);
        const uint32_t* oidx = reinterpret_cast<const uint32_t*>(rec + L.obs_idx);
        const int n_obs = min(h->n_obs, d.max_obs);
        CorbBAEdge* out = FILL ? d.edges + d.edge_off[j] : nullptr;
        const RecLayout KL(d.max_features);
        for (int k = 0; k < n_obs; k++) {
            const int p = corb_idtab_find(d.tab, okf[k]);
            if (p < 0 || d.kf_bad[p]) continue;                            // if(pKF->isBad() || pKF->mnId>maxKFid) continue; (:131-132) -- a keyframe outside the problem
            const char* krec = d.kf_base + (size_t)d.kf_slots[p] * d.kf_bytes;
            const KfHeader* kh = reinterpret_cast<const KfHeader*>(krec);
            const uint32_t f = oidx[k];
            if ((int)f >= kh->n) { if (!FILL) atomicOr(d.status, BAS_BAD_FEATURE); continue; }
            if (FILL) {
                const CorbKeyPoint kp = reinterpret_cast<const CorbKeyPoint*>(krec + KL.kp)[f];       // pKF->mvKeysUn[idx] (rectified stereo: mvKeysUn = mvKeys, Frame.cc:414-420)
                const float ur = reinterpret_cast<const float*>(krec + KL.ur)[f];
                const int oct = min(max(kp.octave, 0), CORB_MAX_LEVELS - 1);
                CorbBAEdge e; e.pose = p; e.point = j; e.u = kp.x; e.v = kp.y; e.u_right = ur;       // mvuRight<0 -> monocular edge (:138)
                e.inv_sigma2 = kh->m.inv_level_sigma2[oct];
                out[cnt] = e;
            }
            cnt++;
        }
    }
    if (!FILL) d.edge_cnt[j] = cnt;
}

// camera centre of vertex p from the solver's output pose (KeyFrame::SetPose: Ow = -Rwc * tcw in float, KeyFrame.cc:120-135)
__device__ __forceinline__ void bas_camera_center(const float* T, float* Ow)
{
#pragma unroll
    for (int a = 0; a < 3; a++) Ow[a] = -(T[0 * 4 + a] * T[3] + T[1 * 4 + a] * T[7] + T[2 * 4 + a] * T[11]);
}
// MapPoint::UpdateNormalAndDepth (C/src/MapPoint.cc:424-472) on a record: the observations whose keyframes are vertices of the problem, with the poses the solve
// left; mvScaleFactors rebuilt from scale_factor as ORBextractor.cc:418-424 does
__device__ __forceinline__ void bas_update_normal_depth(const BAStoreDev& d, CorbMapPointRecord* h, const unsigned long long* okf, const uint32_t* oidx, int kept, const RecLayout& KL, float scale_factor)
{
    const int pr = corb_idtab_find(d.tab, h->ref_kf_id);
    if (pr < 0) return;                                                       // pRefKF == nullptr (:438)
    int ref_f = -1;
    float nx = 0.f, ny = 0.f, nz = 0.f; int n = 0;
    const float px = h->world_pos[0], py = h->world_pos[1], pz = h->world_pos[2];
    for (int k = 0; k < kept; k++) {
        if (okf[k] == h->ref_kf_id) ref_f = (int)oidx[k];
        const int p = corb_idtab_find(d.tab, okf[k]);
        if (p < 0) continue;                                                  // if (pKF) (:452): a keyframe that is not at hand
        float Ow[3]; bas_camera_center(d.poses + 16 * (size_t)p, Ow);
        const float vx = px - Ow[0], vy = py - Ow[1], vz = pz - Ow[2];
        const double inv = 1.0 / sqrt((double)vx * vx + (double)vy * vy + (double)vz * vz);     // cv::norm accumulates in double; Mat / double scales by its reciprocal
        nx += (float)(vx * inv); ny += (float)(vy * inv); nz += (float)(vz * inv); n++;
    }
    const char* rrec = d.kf_base + (size_t)d.kf_slots[pr] * d.kf_bytes;
    const KfHeader* rh = reinterpret_cast<const KfHeader*>(rrec);
    if (ref_f < 0 || ref_f >= rh->n || n == 0) return;                        // (:441-444)
    float Or[3]; bas_camera_center(d.poses + 16 * (size_t)pr, Or);
    const float cx = px - Or[0], cy = py - Or[1], cz = pz - Or[2];
    const float dist = (float)sqrt((double)cx * cx + (double)cy * cy + (double)cz * cz);
    const int nl = min(max(rh->m.nlevels, 1), CORB_MAX_LEVELS);
    const int level = min(max(reinterpret_cast<const CorbKeyPoint*>(rrec + KL.kp)[ref_f].octave, 0), nl - 1);
    float sc = 1.f, sc_level = 1.f;                                           // mvScaleFactor[i] = mvScaleFactor[i-1] * scaleFactor (ORBextractor.cc:418-424)
    for (int l = 1; l < nl; l++) { sc *= scale_factor; if (l == level) sc_level = sc; }
    h->max_distance = dist * sc_level;
    h->min_distance = h->max_distance / sc;
    const double rn = 1.0 / (double)n;
    h->normal[0] = (float)(nx * rn); h->normal[1] = (float)(ny * rn); h->normal[2] = (float)(nz * rn);
}
// estimates -> records (Optimizer.cc:216-262).  poses / points hold the solver's outputs (inputs copied through for fixed / untouched vertices).
// scale_factor > 0 and loop_kf == 0: SetWorldPos is followed by UpdateNormalAndDepth (:254-256) over the point's observers in the solve.
__global__ __launch_bounds__(256) void bas_writeback_kernel(BAStoreDev d, unsigned long long loop_kf, float scale_factor)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d.n_kf) {
        KfHeader* h = reinterpret_cast<KfHeader*>(d.kf_base + (size_t)d.kf_slots[i] * d.kf_bytes);
        // if(pKF->isBad()|| pKF->getFixed()) continue; (:221-222) -- the keyframe with mnId == 1 is written too (its estimate did not move)
        if (!d.kf_bad[i] && !(h->m.flags & CORB_KF_FIXED)) {
            const float* T = d.poses + 16 * (size_t)i;
            float* dst = loop_kf == 0 ? h->m.Tcw : h->m.TcwGBA;           // pKF->SetPose (:228) / pKF->mTcwGBA (:233-235)
#pragma unroll
            for (int k = 0; k < 16; k++) dst[k] = T[k];
            if (loop_kf != 0) h->m.ba_global_for_kf = loop_kf;
        }
    }
    if (i < d.n_mp) {
        CorbMapPointRecord* h = reinterpret_cast<CorbMapPointRecord*>(d.mp_base + (size_t)d.mp_slots[i] * d.mp_bytes);
        // if(vbNotIncludedMP[i]) continue; if(pMP->isBad() || pMP->getFixed()) continue; (:243-248)
        if (!d.mp_bad[i] && !(h->flags & CORB_MP_FIXED) && d.edge_cnt[i] > 0) {
            const float* p = d.points + 3 * (size_t)i;
            float* dst = loop_kf == 0 ? h->world_pos : h->pos_gba;        // pMP->SetWorldPos (:254) / pMP->mPosGBA (:259-261)
            dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
            if (loop_kf != 0) h->ba_global_for_kf = loop_kf;
            else if (scale_factor > 0.f) {
                char* rec = reinterpret_cast<char*>(h);
                const MpLayout L(d.max_obs); const RecLayout KL(d.max_features);
                bas_update_normal_depth(d, h, reinterpret_cast<const unsigned long long*>(rec + L.obs_kf), reinterpret_cast<const uint32_t*>(rec + L.obs_idx), min(h->n_obs, d.max_obs), KL, scale_factor);
            }
        }
    }
}

// ---- the tail of Optimizer::LocalBundleAdjustment (C/src/Optimizer.cc:760-836) on the records ----
// One thread per local map point walks its observation list once more, in the order the fill pass numbered the edges:
//   vToErase (:766-807): pKFi->EraseMapPointMatch(pMP) -> the keyframe record's map-point id of that feature = CORB_NO_MAP_POINT;
//     pMP->EraseObservation(pKFi) (MapPoint.cc:192-217) -> the entry leaves the list, mpRefKF moves to the first remaining observation if it was that keyframe,
//     nObs (2 per stereo observation, 1 per monocular one; an observation whose keyframe is outside the problem counts 1) <= 2 -> SetBadFlag (:255-269):
//     CORB_MP_BAD, list cleared, the matches in its remaining keyframes of the problem cleared
//   pMP->SetWorldPos(estimate) for every local point that is not fixed (:824-830), then pMP->UpdateNormalAndDepth() (MapPoint.cc:424-472) over the
//     observations whose keyframes are vertices, with the poses the solve left
// and one thread per local keyframe writes its pose (pKF->SetPose, :811-820).
__global__ __launch_bounds__(256) void bas_local_finish_kernel(BAStoreDev d, const uint8_t* outlier, int apply_erase, float scale_factor)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d.n_local) {
        KfHeader* h = reinterpret_cast<KfHeader*>(d.kf_base + (size_t)d.kf_slots[i] * d.kf_bytes);
        if (!d.kf_bad[i] && !(h->m.flags & CORB_KF_FIXED)) {                  // if( !pKF->getFixed() ) (:814)
            const float* T = d.poses + 16 * (size_t)i;
#pragma unroll
            for (int k = 0; k < 16; k++) h->m.Tcw[k] = T[k];
        }
    }
    if (i >= d.n_mp) return;
    char* rec = d.mp_base + (size_t)d.mp_slots[i] * d.mp_bytes;
    CorbMapPointRecord* h = reinterpret_cast<CorbMapPointRecord*>(rec);
    if (d.mp_bad[i]) return;                                                  // (lLocalMapPoints holds no bad point, :518)
    const MpLayout L(d.max_obs);
    const RecLayout KL(d.max_features);
    unsigned long long* okf = reinterpret_cast<unsigned long long*>(rec + L.obs_kf);
    uint32_t* oidx = reinterpret_cast<uint32_t*>(rec + L.obs_idx);
    const int n_obs = min(h->n_obs, d.max_obs);
    int e = d.edge_off[i], kept = 0, weight = 0;
    bool ref_erased = false;
    for (int k = 0; k < n_obs; k++) {
        const unsigned long long id = okf[k]; const uint32_t f = oidx[k];
        const int p = corb_idtab_find(d.tab, id);
        int w = 1; bool erase = false;
        if (p >= 0) {
            char* krec = d.kf_base + (size_t)d.kf_slots[p] * d.kf_bytes;
            if ((int)f < reinterpret_cast<const KfHeader*>(krec)->n) {
                if (!d.kf_bad[p]) { erase = apply_erase && outlier[e] != 0; e++; }      // (the same three tests as the fill pass: this observation is edge e)
                if (erase) reinterpret_cast<unsigned long long*>(krec + KL.mp_id)[f] = CORB_NO_MAP_POINT;
                else w = reinterpret_cast<const float*>(krec + KL.ur)[f] >= 0.f ? 2 : 1;      // (nObs is the point's running count: a bad keyframe of the problem has no edge, but its observation weighs what it weighed when it was added)
            }
        }
        if (erase) { if (id == h->ref_kf_id) ref_erased = true; continue; }
        okf[kept] = id; oidx[kept] = f; kept++; weight += w;
    }
    const bool fixed = (h->flags & CORB_MP_FIXED) != 0;
    if (!fixed) { const float* q = d.points + 3 * (size_t)i; h->world_pos[0] = q[0]; h->world_pos[1] = q[1]; h->world_pos[2] = q[2]; }
    if (kept != n_obs) {
        h->n_obs = kept;
        if (ref_erased && kept > 0) h->ref_kf_id = okf[0];
        if (weight <= 2) {                                                    // SetBadFlag
            for (int k = 0; k < kept; k++) {
                const int p = corb_idtab_find(d.tab, okf[k]);
                if (p < 0) continue;
                char* krec = d.kf_base + (size_t)d.kf_slots[p] * d.kf_bytes;
                if ((int)oidx[k] < reinterpret_cast<const KfHeader*>(krec)->n) reinterpret_cast<unsigned long long*>(krec + KL.mp_id)[oidx[k]] = CORB_NO_MAP_POINT;
            }
            h->flags |= CORB_MP_BAD; h->n_obs = 0;
            return;
        }
    }
    if (fixed || kept == 0) return;
    bas_update_normal_depth(d, h, okf, oidx, kept, KL, scale_factor);
}

void bas_launch_vertices(const BAStoreDev& d, hipStream_t s)
{
    if (d.n_kf > 0) hipLaunchKernelGGL(bas_kf_kernel, dim3((d.n_kf + 255) / 256), dim3(256), 0, s, d);
}
void bas_launch_count(const BAStoreDev& d, hipStream_t s)
{
    if (d.n_mp > 0) hipLaunchKernelGGL(bas_mp_kernel<false>, dim3((d.n_mp + 255) / 256), dim3(256), 0, s, d);
}
void bas_launch_fill(const BAStoreDev& d, hipStream_t s)
{
    if (d.n_mp > 0) hipLaunchKernelGGL(bas_mp_kernel<true>, dim3((d.n_mp + 255) / 256), dim3(256), 0, s, d);
}
void bas_launch_writeback(const BAStoreDev& d, unsigned long long loop_kf, float scale_factor, hipStream_t s)
{
    const int n = d.n_kf > d.n_mp ? d.n_kf : d.n_mp;
    if (n > 0) hipLaunchKernelGGL(bas_writeback_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, loop_kf, scale_factor);
}
void bas_launch_local_finish(const BAStoreDev& d, const uint8_t* edge_outlier, int apply_erase, float scale_factor, hipStream_t s)
{
    const int n = d.n_local > d.n_mp ? d.n_local : d.n_mp;
    if (n > 0) hipLaunchKernelGGL(bas_local_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, edge_outlier, apply_erase, scale_factor);
}

__global__ __launch_bounds__(1024) void bas_local_results_kernel(BAStoreDev d, const uint8_t* __restrict__ outlier, int n_edges, int* __restrict__ block, int pairs_off, int poses_off, int points_off)
{
    __shared__ int sh[16];
    const int t = threadIdx.x;
    const float* P = d.poses; const float* X = d.points;
    float* bp = reinterpret_cast<float*>(block + poses_off); float* bx = reinterpret_cast<float*>(block + points_off);
    for (int i = t; i < 16 * d.n_kf; i += 1024) bp[i] = P[i];
    for (int i = t; i < 3 * d.n_mp; i += 1024) bx[i] = X[i];
    // ordered compaction: a thread owns ceil(n_edges / 1024) consecutive edges
    const int per = (n_edges + 1023) / 1024;
    const int b = min(t * per, n_edges), e = min(b + per, n_edges);
    int cnt = 0;
    for (int i = b; i < e; i++) cnt += outlier[i] ? 1 : 0;
    const int lane = t & 63, w = t >> 6;
    int inc = cnt;
    inc = lx_wave_incl_scan_i(inc);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    int base = 0, total = 0;
    for (int i = 0; i < 16; i++) { if (i < w) base += sh[i]; total += sh[i]; }
    int at = base + inc - cnt;
    int2* pairs = reinterpret_cast<int2*>(block + pairs_off);
    for (int i = b; i < e; i++) if (outlier[i]) { pairs[at] = make_int2(d.edges[i].pose, d.edges[i].point); at++; }
    if (t == 0) block[0] = total;
}
void bas_launch_local_results(const BAStoreDev& d, const uint8_t* edge_outlier, int n_edges, int* block, int pairs_off, int poses_off, int points_off, hipStream_t s)
{
    hipLaunchKernelGGL(bas_local_results_kernel, dim3(1), dim3(1024), 0, s, d, edge_outlier, n_edges, block, pairs_off, poses_off, points_off);
}
