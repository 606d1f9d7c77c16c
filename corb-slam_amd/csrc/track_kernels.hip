// track_kernels.hip -- gathers / scatters between store records and the flat views of the projection matcher and the single-pose optimiser
// (Tracking::TrackWithMotionModel, C/src/Tracking.cc:868-940: SearchByProjection(CurrentFrame, LastFrame, th, mono) -> PoseOptimization(&CurrentFrame)).
#include "track_internal.h"
#include "ba_math.h"

__global__ __launch_bounds__(256) void track_index_kernel(const char* mp_base, size_t mp_bytes, int first, int n, CorbIdTable idt, int* dup)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const CorbMapPointRecord* r = reinterpret_cast<const CorbMapPointRecord*>(mp_base + (size_t)(first + i) * mp_bytes);
    if (!corb_idtab_insert(idt, r->id, first + i)) *dup = 1;
}
void track_launch_index_store(const char* mp_base, size_t mp_bytes, int first, int n, CorbIdTable idt, int* dup, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(track_index_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mp_base, mp_bytes, first, n, idt, dup);
}

__device__ __forceinline__ const CorbMapPointRecord* track_find_mp(const char* mp_base, size_t mp_bytes, const CorbIdTable& idt, unsigned long long id)
{
    if (id == CORB_NO_MAP_POINT) return nullptr;
    const int slot = corb_idtab_find(idt, id);
    return slot < 0 ? nullptr : reinterpret_cast<const CorbMapPointRecord*>(mp_base + (size_t)slot * mp_bytes);
}

// the MapPoint id a feature HOLDS (discarded ones hold none)
__device__ __forceinline__ unsigned long long track_held_id(const char* rec, const RecLayout& L, int i)
{
    const unsigned char fl = reinterpret_cast<const unsigned char*>(rec + L.flags)[i];
    return (fl & CORB_FEATURE_DISCARDED) ? CORB_NO_MAP_POINT : reinterpret_cast<const unsigned long long*>(rec + L.mp_id)[i];
}

__global__ __launch_bounds__(256) void track_prepare_last_kernel(TrackDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const RecLayout L(t.F);
    if (i < t.n_last) {
        // LastFrame.mvpMapPoints[i] && !LastFrame.mvbOutlier[i] (ORBmatcher.cc:1496-1500); a MapPoint that is not in the store or is bad is no MapPoint
        const unsigned long long id = track_held_id(t.last, L, i);
        const CorbMapPointRecord* r = track_find_mp(t.mp_base, t.mp_bytes, t.idt, id);
        const CorbKeyPoint k = reinterpret_cast<const CorbKeyPoint*>(t.last + L.kp)[i];
        const unsigned char fl = reinterpret_cast<const unsigned char*>(t.last + L.flags)[i];
        CorbLastPoint o; o.world[0] = o.world[1] = o.world[2] = 0.f; o.angle = k.angle; o.octave = k.octave; o.valid = 0; o.claims = 0; o.pad[0] = o.pad[1] = 0;
        unsigned long long dsc[4] = {0, 0, 0, 0};
        if (r && !(r->flags & CORB_MP_BAD) && !(fl & CORB_FEATURE_OUTLIER)) {
            o.world[0] = r->world_pos[0]; o.world[1] = r->world_pos[1]; o.world[2] = r->world_pos[2];
            o.valid = 1; o.claims = r->n_obs > 0 ? 1 : 0;
            const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(r->descriptor);     // (byte offset 16 of a 64-byte aligned record: aligned 64-bit loads, see the static_assert in store_internal.h)
            dsc[0] = dp[0]; dsc[1] = dp[1]; dsc[2] = dp[2]; dsc[3] = dp[3];
        }
        t.lastp[i] = o;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) t.qdesc[4 * (size_t)i + k4] = dsc[k4];
    }
    if (i < t.n_cur) {
        // if(CurrentFrame.mvpMapPoints[i2]) if(CurrentFrame.mvpMapPoints[i2]->Observations()>0) continue;  (ORBmatcher.cc:1545-1547)
        const CorbMapPointRecord* r = track_find_mp(t.mp_base, t.mp_bytes, t.idt, track_held_id(t.cur, L, i));
        t.claimed[i] = (r && r->n_obs > 0) ? 1 : 0;
    }
}
void track_launch_prepare_last(const TrackDev& t, hipStream_t s)
{
    const int n = t.n_cur > t.n_last ? t.n_cur : t.n_last;
    if (n > 0) hipLaunchKernelGGL(track_prepare_last_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t);
}

__global__ __launch_bounds__(256) void track_scatter_last_kernel(TrackDev t)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= t.n_cur) return;
    const int m = t.match[f];
    if (m < 0) return;
    const RecLayout L(t.F);
    reinterpret_cast<unsigned long long*>(t.cur + L.mp_id)[f] = reinterpret_cast<const unsigned long long*>(t.last + L.mp_id)[m];
    reinterpret_cast<unsigned char*>(t.cur + L.flags)[f] &= (unsigned char)~(CORB_FEATURE_DISCARDED | CORB_FEATURE_OUTLIER);
}
void track_launch_scatter_last(const TrackDev& t, hipStream_t s)
{
    if (t.n_cur > 0) hipLaunchKernelGGL(track_scatter_last_kernel, dim3((t.n_cur + 255) / 256), dim3(256), 0, s, t);
}

// ---- PoseOptimization(Frame*) on a record ----
// one workgroup: ordered compaction of the features that hold a usable MapPoint (the reference adds its edges for i = 0 .. N-1, Optimizer.cc:300-366)
__global__ __launch_bounds__(1024) void track_pose_gather_kernel(TrackPoseDev t)
{
    __shared__ int wsum[16], base;
    const RecLayout L(t.F);
    const KfHeader* H = reinterpret_cast<const KfHeader*>(t.cur);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < t.n_cur; i0 += 1024) {
        const int i = i0 + tid;
        const CorbMapPointRecord* r = nullptr;
        if (i < t.n_cur) {
            r = track_find_mp(t.mp_base, t.mp_bytes, t.idt, track_held_id(t.cur, L, i));
            if (r && (r->flags & CORB_MP_BAD)) r = nullptr;
        }
        const unsigned long long m = __ballot(r != nullptr);
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; w++) before += wsum[w];
        if (r) {
            const int e = before + __popcll(m & ((1ull << lane) - 1ull));
            const CorbKeyPoint k = reinterpret_cast<const CorbKeyPoint*>(t.cur + L.kp)[i];
            const float ur = reinterpret_cast<const float*>(t.cur + L.ur)[i];
            t.pt[3 * (size_t)e] = (double)r->world_pos[0]; t.pt[3 * (size_t)e + 1] = (double)r->world_pos[1]; t.pt[3 * (size_t)e + 2] = (double)r->world_pos[2];
            t.obs[3 * (size_t)e] = (double)k.x; t.obs[3 * (size_t)e + 1] = (double)k.y; t.obs[3 * (size_t)e + 2] = (double)ur;
            t.w[e] = (double)H->m.inv_level_sigma2[k.octave];
            t.dim[e] = ur < 0 ? 2 : 3;                             // mvuRight < 0 -> monocular edge (Optimizer.cc:310)
            t.efeat[e] = i;
        }
        __syncthreads();
        if (tid == 0) { int tot = 0; for (int w = 0; w < 16; w++) tot += wsum[w]; base += tot; }
        __syncthreads();
    }
    if (tid == 0) { t.edge_off[0] = 0; t.edge_off[1] = base; t.stage_limit[0] = base < 3 ? 0 : base < 10 ? 1 : 4; }
}
void track_launch_pose_gather(const TrackPoseDev& t, hipStream_t s) { hipLaunchKernelGGL(track_pose_gather_kernel, dim3(1), dim3(1024), 0, s, t); }

__global__ __launch_bounds__(256) void track_pose_finish_kernel(TrackPoseDev t)          // ONE workgroup
{
    const RecLayout L(t.F);
    unsigned char* fl = reinterpret_cast<unsigned char*>(t.cur + L.flags);
    const int E = t.edge_off[1];
    for (int i = threadIdx.x; i < t.n_cur; i += 256) fl[i] &= (unsigned char)~CORB_FEATURE_OUTLIER;       // mvbOutlier[i] = false where no edge says otherwise
    __syncthreads();
    // (one edge per feature: own byte)  discard: mvpMapPoints[i] = NULL, mvbOutlier[i] = false, pMP->mnLastFrameSeen = this frame (Tracking.cc:927-936)
    for (int e = threadIdx.x; e < E; e += 256) if (!t.active[e]) fl[t.efeat[e]] |= (unsigned char)(t.discard ? CORB_FEATURE_DISCARDED : CORB_FEATURE_OUTLIER);
    if (threadIdx.x < 7 && t.result) t.result[threadIdx.x] = t.pose[threadIdx.x];
    if (threadIdx.x >= 8 && threadIdx.x < 14 && t.result) { const int k = threadIdx.x - 8; reinterpret_cast<int*>(t.result + 7)[k] = k < 4 ? t.counters[k] : t.edge_off[k - 4]; }
    if (threadIdx.x == 0 && t.counters[2]) {                                                             // the graph had an active edge: pFrame->SetPose (Optimizer.cc:478-481)
        KfHeader* H = reinterpret_cast<KfHeader*>(t.cur);
        double R[9]; quat_to_R(t.pose, R);
        float* T = H->m.Tcw;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T[r * 4 + c] = (float)R[r * 3 + c]; T[r * 4 + 3] = (float)t.pose[4 + r]; }
        T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
    }
}
void track_launch_pose_finish(const TrackPoseDev& t, hipStream_t s) { hipLaunchKernelGGL(track_pose_finish_kernel, dim3(1), dim3(256), 0, s, t); }

// ---- Tracking::SearchLocalPoints on a record ----
__global__ __launch_bounds__(256) void track_local_frame_kernel(TrackLocalDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n_cur) return;
    const RecLayout L(t.F);
    unsigned long long* mid = reinterpret_cast<unsigned long long*>(t.cur + L.mp_id);
    const unsigned long long id = mid[i];
    const bool discarded = reinterpret_cast<const unsigned char*>(t.cur + L.flags)[i] & CORB_FEATURE_DISCARDED;
    const CorbMapPointRecord* r = track_find_mp(t.mp_base, t.mp_bytes, t.idt, id);
    unsigned char cl = 0;
    if (r) {
        if (discarded) (void)corb_idtab_insert(t.inframe, id, i);          // pMP->mnLastFrameSeen = mCurrentFrame.mnId of a discarded outlier (:933)
        else if (r->flags & CORB_MP_BAD) mid[i] = CORB_NO_MAP_POINT;       // if(pMP->isBad()) *vit = NULL;  (Tracking.cc:1176-1179)
        else { (void)corb_idtab_insert(t.inframe, id, i); cl = r->n_obs > 0 ? 1 : 0; }     // pMP->mnLastFrameSeen = mCurrentFrame.mnId (:1183)
    }
    t.claimed[i] = cl;
}
// (3x3 products = cv::gemm on CV_32F: double accumulation, one rounding; cv::norm / Mat::dot = double sums; PredictScale's log as in proj_kernels.hip)
__global__ __launch_bounds__(256) void track_local_points_kernel(TrackLocalDev t)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= t.n_local) return;
    CorbTrackedPoint o; o.proj_x = o.proj_y = o.proj_xr = o.view_cos = 0.f; o.level = 0; o.valid = 0; o.claims = 0; o.pad[0] = o.pad[1] = 0;
    unsigned long long dsc[4] = {0, 0, 0, 0};
    const unsigned long long id = t.ids[q];
    const CorbMapPointRecord* r = track_find_mp(t.mp_base, t.mp_bytes, t.idt, id);
    if (r && !(r->flags & CORB_MP_BAD) && corb_idtab_find(t.inframe, id) < 0) {
        const float* P = r->world_pos;
        float Pc[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double s = __fma_rn((double)t.Tcw[i * 4 + 2], (double)P[2], __fma_rn((double)t.Tcw[i * 4 + 1], (double)P[1], __dmul_rn((double)t.Tcw[i * 4], (double)P[0])));
            Pc[i] = (float)__dadd_rn(s, (double)t.Tcw[i * 4 + 3]);
        }
        if (Pc[2] > 0.0f) {
            const float invz = __fdiv_rn(1.0f, Pc[2]);
            const float u = __fadd_rn(__fmul_rn(__fmul_rn(t.fx, Pc[0]), invz), t.cx), v = __fadd_rn(__fmul_rn(__fmul_rn(t.fy, Pc[1]), invz), t.cy);
            if (!(u < t.min_x || u > t.max_x || v < t.min_y || v > t.max_y)) {
                const float maxD = __fmul_rn(1.2f, r->max_distance), minD = __fmul_rn(0.8f, r->min_distance);
                const float PO[3] = { __fsub_rn(P[0], t.Ow[0]), __fsub_rn(P[1], t.Ow[1]), __fsub_rn(P[2], t.Ow[2]) };
                const float dist = (float)sqrt(__fma_rn((double)PO[2], (double)PO[2], __fma_rn((double)PO[1], (double)PO[1], __dmul_rn((double)PO[0], (double)PO[0]))));
                if (!(dist < minD || dist > maxD)) {
                    const double dot = __fma_rn((double)PO[2], (double)r->normal[2], __fma_rn((double)PO[1], (double)r->normal[1], __dmul_rn((double)PO[0], (double)r->normal[0])));
                    const float viewCos = (float)__ddiv_rn(dot, (double)dist);
                    if (!(viewCos < t.cos_limit)) {
                        const float ratio = __fdiv_rn(r->max_distance, dist);
                        const float lg = (float)log((double)ratio);
                        int lvl = (int)ceilf(__fdiv_rn(lg, t.log_scale));
                        if (lvl < 0) lvl = 0; else if (lvl >= t.nlevels) lvl = t.nlevels - 1;
                        o.proj_x = u; o.proj_y = v; o.proj_xr = __fsub_rn(u, __fmul_rn(t.bf, invz)); o.view_cos = viewCos; o.level = lvl;
                        o.valid = 1; o.claims = r->n_obs > 0 ? 1 : 0;
                        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(r->descriptor);
                        dsc[0] = dp[0]; dsc[1] = dp[1]; dsc[2] = dp[2]; dsc[3] = dp[3];
                        atomicAdd(t.n_in_view, 1);
                    }
                }
            }
        }
    }
    t.tracked[q] = o;
#pragma unroll
    for (int k = 0; k < 4; k++) t.qdesc[4 * (size_t)q + k] = dsc[k];
}
void track_launch_prepare_local(const TrackLocalDev& t, hipStream_t s)
{
    if (t.n_cur > 0) hipLaunchKernelGGL(track_local_frame_kernel, dim3((t.n_cur + 255) / 256), dim3(256), 0, s, t);
    if (t.n_local > 0) hipLaunchKernelGGL(track_local_points_kernel, dim3((t.n_local + 255) / 256), dim3(256), 0, s, t);
}
__global__ __launch_bounds__(256) void track_scatter_local_kernel(TrackLocalDev t)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= t.n_cur) return;
    const int m = t.match[f];
    if (m < 0) return;
    const RecLayout L(t.F);
    reinterpret_cast<unsigned long long*>(t.cur + L.mp_id)[f] = t.ids[m];
    reinterpret_cast<unsigned char*>(t.cur + L.flags)[f] &= (unsigned char)~(CORB_FEATURE_DISCARDED | CORB_FEATURE_OUTLIER);
}
void track_launch_scatter_local(const TrackLocalDev& t, hipStream_t s)
{
    if (t.n_cur > 0) hipLaunchKernelGGL(track_scatter_local_kernel, dim3((t.n_cur + 255) / 256), dim3(256), 0, s, t);
}

// ---- ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) on records (C/src/ORBmatcher.cc:960-1116) ----
__global__ __launch_bounds__(256) void fuse_prepare_kernel(FuseStoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < t.n_feat) t.claim[i] = 0x7FFFFFFF;
    if (i >= t.n_points) return;
    const char* rec = t.mp_base + (size_t)t.mp_slots[i] * t.mp_bytes;
    const CorbMapPointRecord* h = reinterpret_cast<const CorbMapPointRecord*>(rec);
    const unsigned long long kf_id = reinterpret_cast<const KfHeader*>(t.kf_rec)->m.id;
    bool ok = (h->flags & CORB_MP_BAD) == 0;                     // if(pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue; (:990-993)
    if (ok) {
        const MpLayout L(t.max_obs);
        const unsigned long long* okf = reinterpret_cast<const unsigned long long*>(rec + L.obs_kf);
        const int n_obs = min(h->n_obs, t.max_obs);
        for (int k = 0; k < n_obs; k++) if (okf[k] == kf_id) { ok = false; break; }
    }
    CorbMapPointView v;
    v.world[0] = h->world_pos[0]; v.world[1] = h->world_pos[1]; v.world[2] = h->world_pos[2];
    v.normal[0] = h->normal[0]; v.normal[1] = h->normal[1]; v.normal[2] = h->normal[2];
    v.min_distance = h->min_distance; v.max_distance = h->max_distance; v.angle = 0.f; v.valid = ok ? 1 : 0; v.pad[0] = v.pad[1] = v.pad[2] = 0;
    t.pts[i] = v;
    const unsigned long long* dsc = reinterpret_cast<const unsigned long long*>(h->descriptor);
#pragma unroll
    for (int k = 0; k < 4; k++) t.qdesc[4 * (size_t)i + k] = dsc[k];
}
__global__ __launch_bounds__(256) void fuse_claim_kernel(FuseStoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n_points) return;
    const int f = t.best_idx[i];
    if (f >= 0) atomicMin(&t.claim[f], i);                       // vpMapPoints is walked in order: the first point fused into a feature finds it as it was
}
// The reference walks the points in order; a fused point whose feature holds no MapPoint enters it (pMP->AddObservation(pKF,bestIdx); pKF->AddMapPoint(pMP,bestIdx),
// :1097-1101) -- every later point fused into the same feature, and every point whose feature held a MapPoint before the call, meets a MapPoint there and ends in
// MapPoint::Replace (:1085-1096), which re-links whole observation lists: reported (action 2), left to the caller.
__global__ __launch_bounds__(256) void fuse_apply_kernel(FuseStoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n_points) return;
    const int f = t.best_idx[i];
    unsigned char act = 0;
    if (f >= 0) {
        const RecLayout KL(t.F);
        unsigned long long* mp_id = reinterpret_cast<unsigned long long*>(t.kf_rec + KL.mp_id);
        if (t.claim[f] != i || mp_id[f] != CORB_NO_MAP_POINT) act = 2;          // (only the claiming point writes mp_id[f]: it reads the value from before the call)
        else {
            act = 1;
            if (t.apply) {
                char* rec = t.mp_base + (size_t)t.mp_slots[i] * t.mp_bytes;
                CorbMapPointRecord* h = reinterpret_cast<CorbMapPointRecord*>(rec);
                const MpLayout L(t.max_obs);
                unsigned long long* okf = reinterpret_cast<unsigned long long*>(rec + L.obs_kf);
                uint32_t* oidx = reinterpret_cast<uint32_t*>(rec + L.obs_idx);
                const unsigned long long kf_id = reinterpret_cast<const KfHeader*>(t.kf_rec)->m.id;
                const int n = h->n_obs;
                if (n >= t.max_obs) act = 3;
                else {                                                           // mObservations[pKF] = idx: the list ascends in the keyframe id (MapPoint.h:182)
                    int k = n;
                    while (k > 0 && okf[k - 1] > kf_id) { okf[k] = okf[k - 1]; oidx[k] = oidx[k - 1]; k--; }
                    okf[k] = kf_id; oidx[k] = (uint32_t)f; h->n_obs = n + 1;
                    mp_id[f] = h->id;
                }
            }
        }
    }
    t.action[i] = act;
}
void fuse_launch_prepare(const FuseStoreDev& t, hipStream_t s)
{
    const int n = t.n_points > t.n_feat ? t.n_points : t.n_feat;
    if (n > 0) hipLaunchKernelGGL(fuse_prepare_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t);
}
void fuse_launch_apply(const FuseStoreDev& t, hipStream_t s)
{
    if (t.n_points <= 0) return;
    hipLaunchKernelGGL(fuse_claim_kernel, dim3((t.n_points + 255) / 256), dim3(256), 0, s, t);
    hipLaunchKernelGGL(fuse_apply_kernel, dim3((t.n_points + 255) / 256), dim3(256), 0, s, t);
}

// ---- ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) on records (C/src/ORBmatcher.cc:1616-1744; Tracking::Relocalization,
// C/src/Tracking.cc:1440-1500, calls it with sAlreadyFound = the MapPoints the frame holds) ----
__global__ __launch_bounds__(256) void reloc_frame_kernel(RelocStoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n_cur) return;
    const RecLayout L(t.F_cur);
    const unsigned long long id = track_held_id(t.cur, L, i);
    // if(CurrentFrame.mvpMapPoints[i2]) continue; (:1680): a held pointer counts whether or not the point is bad or known to this store
    t.claimed[i] = id != CORB_NO_MAP_POINT ? 1 : 0;
    if (id != CORB_NO_MAP_POINT) (void)corb_idtab_insert(t.inframe, id, i);
}
__global__ __launch_bounds__(256) void reloc_points_kernel(RelocStoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n_kf) return;
    const RecLayout L(t.F_kf);
    const unsigned long long id = reinterpret_cast<const unsigned long long*>(t.kf + L.mp_id)[i];      // pKF->GetMapPointMatches()[i]
    const CorbMapPointRecord* r = track_find_mp(t.mp_base, t.mp_bytes, t.idt, id);
    CorbMapPointView v; memset(&v, 0, sizeof(v));
    unsigned long long dsc[4] = {0, 0, 0, 0};
    if (r && !(r->flags & CORB_MP_BAD) && corb_idtab_find(t.inframe, id) < 0) {                        // pMP && !pMP->isBad() && !sAlreadyFound.count(pMP) (:1636-1639)
        v.world[0] = r->world_pos[0]; v.world[1] = r->world_pos[1]; v.world[2] = r->world_pos[2];
        v.normal[0] = r->normal[0]; v.normal[1] = r->normal[1]; v.normal[2] = r->normal[2];
        v.min_distance = r->min_distance; v.max_distance = r->max_distance;
        v.angle = reinterpret_cast<const CorbKeyPoint*>(t.kf + L.kp)[i].angle;                          // pKF->mvKeysUn[i].angle (:1699)
        v.valid = 1;
        const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(r->descriptor);
        dsc[0] = dp[0]; dsc[1] = dp[1]; dsc[2] = dp[2]; dsc[3] = dp[3];
    }
    t.pts[i] = v;
#pragma unroll
    for (int k = 0; k < 4; k++) t.qdesc[4 * (size_t)i + k] = dsc[k];
}
// CurrentFrame.mvpMapPoints[bestIdx2] = pMP (:1692)
__global__ __launch_bounds__(256) void reloc_scatter_kernel(RelocStoreDev t)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= t.n_cur) return;
    const int m = t.match[f];
    if (m < 0) return;
    const RecLayout L(t.F_cur), LK(t.F_kf);
    reinterpret_cast<unsigned long long*>(t.cur + L.mp_id)[f] = reinterpret_cast<const unsigned long long*>(t.kf + LK.mp_id)[m];
    reinterpret_cast<unsigned char*>(t.cur + L.flags)[f] &= (unsigned char)~(CORB_FEATURE_DISCARDED | CORB_FEATURE_OUTLIER);
}
void reloc_launch_prepare(const RelocStoreDev& t, hipStream_t s)
{
    if (t.n_cur > 0) hipLaunchKernelGGL(reloc_frame_kernel, dim3((t.n_cur + 255) / 256), dim3(256), 0, s, t);
    if (t.n_kf > 0) hipLaunchKernelGGL(reloc_points_kernel, dim3((t.n_kf + 255) / 256), dim3(256), 0, s, t);
}
void reloc_launch_scatter(const RelocStoreDev& t, hipStream_t s)
{
    if (t.n_cur > 0) hipLaunchKernelGGL(reloc_scatter_kernel, dim3((t.n_cur + 255) / 256), dim3(256), 0, s, t);
}

// ---- ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) on records (C/src/ORBmatcher.cc:425-538; LoopClosing.cc:377,
// S/src/GlobalOptimize.cpp:199): vpMatched travels as MapPoint ids per feature of pKF ----
__global__ __launch_bounds__(256) void scw_feat_kernel(ScwStoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n_feat) return;
    const unsigned long long id = t.matched[i];
    t.claimed[i] = id != CORB_NO_MAP_POINT ? 1 : 0;              // if(vpMatched[idx]) continue; (:510)
    if (id != CORB_NO_MAP_POINT) (void)corb_idtab_insert(t.found, id, i);       // spAlreadyFound(vpMatched.begin(), vpMatched.end()) minus NULL (:441-442)
}
__global__ __launch_bounds__(256) void scw_points_kernel(ScwStoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n_points) return;
    const CorbMapPointRecord* r = reinterpret_cast<const CorbMapPointRecord*>(t.mp_base + (size_t)t.mp_slots[i] * t.mp_bytes);
    CorbMapPointView v; memset(&v, 0, sizeof(v));
    v.world[0] = r->world_pos[0]; v.world[1] = r->world_pos[1]; v.world[2] = r->world_pos[2];
    v.normal[0] = r->normal[0]; v.normal[1] = r->normal[1]; v.normal[2] = r->normal[2];
    v.min_distance = r->min_distance; v.max_distance = r->max_distance;
    v.valid = (!(r->flags & CORB_MP_BAD) && corb_idtab_find(t.found, r->id) < 0) ? 1 : 0;       // if(pMP->isBad() || spAlreadyFound.count(pMP)) continue; (:452)
    t.pts[i] = v;
    const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(r->descriptor);
#pragma unroll
    for (int k = 0; k < 4; k++) t.qdesc[4 * (size_t)i + k] = dp[k];
}
// vpMatched[bestIdx] = pMP (:530)
__global__ __launch_bounds__(256) void scw_scatter_kernel(ScwStoreDev t)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= t.n_feat) return;
    const int m = t.match[f];
    if (m < 0) return;
    t.matched[f] = reinterpret_cast<const CorbMapPointRecord*>(t.mp_base + (size_t)t.mp_slots[m] * t.mp_bytes)->id;
}
void scw_launch_prepare(const ScwStoreDev& t, hipStream_t s)
{
    if (t.n_feat > 0) hipLaunchKernelGGL(scw_feat_kernel, dim3((t.n_feat + 255) / 256), dim3(256), 0, s, t);
    if (t.n_points > 0) hipLaunchKernelGGL(scw_points_kernel, dim3((t.n_points + 255) / 256), dim3(256), 0, s, t);
}
void scw_launch_scatter(const ScwStoreDev& t, hipStream_t s)
{
    if (t.n_feat > 0) hipLaunchKernelGGL(scw_scatter_kernel, dim3((t.n_feat + 255) / 256), dim3(256), 0, s, t);
}

// ---- ORBmatcher::SearchBySim3 on records (C/src/ORBmatcher.cc:1244-1468): the MapPoint views of both keyframes' features ----
// pass 0: vbAlreadyMatched1 / vbAlreadyMatched2 (:1270-1283) from vpMatches12 given as MapPoint ids; pass 1: the views
__global__ __launch_bounds__(256) void sim3_flags_kernel(Sim3StoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= t.n1) return;
    const unsigned long long mid = t.matched12 ? t.matched12[i] : CORB_NO_MAP_POINT;
    t.already1[i] = mid != CORB_NO_MAP_POINT ? 1 : 0;
    if (mid == CORB_NO_MAP_POINT) return;
    // int idx2 = pMP->GetIndexInKeyFrame(pKF2): the observation list of the matched point, entry of keyframe 2
    const int slot = corb_idtab_find(t.idt, mid);
    if (slot < 0) return;
    const char* rec = t.mp_base + (size_t)slot * t.mp_bytes;
    const CorbMapPointRecord* h = reinterpret_cast<const CorbMapPointRecord*>(rec);
    const MpLayout L(t.max_obs);
    const unsigned long long* okf = reinterpret_cast<const unsigned long long*>(rec + L.obs_kf);
    const uint32_t* oidx = reinterpret_cast<const uint32_t*>(rec + L.obs_idx);
    const unsigned long long kf2_id = reinterpret_cast<const KfHeader*>(t.kf2)->m.id;
    const int n_obs = min(h->n_obs, t.max_obs);
    for (int k = 0; k < n_obs; k++) if (okf[k] == kf2_id) { const int idx2 = (int)oidx[k]; if (idx2 >= 0 && idx2 < t.n2) t.already2[idx2] = 1; break; }
}
__global__ __launch_bounds__(256) void sim3_views_kernel(Sim3StoreDev t)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const RecLayout L(t.F);
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const int n = side ? t.n2 : t.n1;
        if (i >= n) continue;
        const char* kf = side ? t.kf2 : t.kf1;
        const unsigned long long id = reinterpret_cast<const unsigned long long*>(kf + L.mp_id)[i];
        const CorbMapPointRecord* r = track_find_mp(t.mp_base, t.mp_bytes, t.idt, id);
        const unsigned char already = side ? t.already2[i] : t.already1[i];
        CorbMapPointView v; memset(&v, 0, sizeof(v));
        unsigned long long dsc[4] = {0, 0, 0, 0};
        if (r && !already && !(r->flags & CORB_MP_BAD)) {                                               // (:1291-1296, :1371-1376)
            v.world[0] = r->world_pos[0]; v.world[1] = r->world_pos[1]; v.world[2] = r->world_pos[2];
            v.normal[0] = r->normal[0]; v.normal[1] = r->normal[1]; v.normal[2] = r->normal[2];
            v.min_distance = r->min_distance; v.max_distance = r->max_distance; v.valid = 1;
            const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(r->descriptor);
            dsc[0] = dp[0]; dsc[1] = dp[1]; dsc[2] = dp[2]; dsc[3] = dp[3];
        }
        (side ? t.pts2 : t.pts1)[i] = v;
        unsigned long long* q = side ? t.qdesc2 : t.qdesc1;
#pragma unroll
        for (int k = 0; k < 4; k++) q[4 * (size_t)i + k] = dsc[k];
    }
}
void sim3_launch_prepare(const Sim3StoreDev& t, hipStream_t s)
{
    if (t.n1 > 0) hipLaunchKernelGGL(sim3_flags_kernel, dim3((t.n1 + 255) / 256), dim3(256), 0, s, t);
    const int n = t.n1 > t.n2 ? t.n1 : t.n2;
    if (n > 0) hipLaunchKernelGGL(sim3_views_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t);
}
