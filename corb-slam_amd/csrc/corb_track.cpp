// corb_track.cpp -- C-ABI host side of the tracking-thread calls on device-resident records (include/corb_accel.h, last section): no feature, descriptor or
// map point crosses PCIe; the host contributes the two poses, the camera and the launch sizes it already knows (the stores' feature counts).
#include "track_internal.h"
#include "store_host.h"
#include "corb_workspace.h"
#include <cstring>
#include <vector>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);
void corb_pose_from_T(const float* T, double* out7);
void corb_pose_to_T(const double* p7, float* T);
void corb_pose_optimization_stages(CorbBAStage* st);

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

extern "C" int corb_mp_store_build_index(CorbMpStore* s, int first, int n)
{
    if (!s || first < 0 || n < 0 || (long long)first + n > s->capacity) { corb_set_error("corb_mp_store_build_index: bad store / slot range"); return CORB_ERR_ARG; }
    int rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    unsigned int cap = 64; while (cap < 2u * (unsigned int)(n > 0 ? n : 1)) cap <<= 1;
    if (!s->idt.keys || s->idt.mask + 1 != cap) {
        if (s->idt.keys) { (void)hipFree(s->idt.keys); s->idt.keys = nullptr; s->idt.vals = nullptr; }
        char* mem = nullptr;
        HIPCHK(hipMalloc((void**)&mem, (size_t)cap * 12 + 256));
        s->idt.keys = reinterpret_cast<unsigned long long*>(mem); s->idt.vals = reinterpret_cast<int*>(mem + (size_t)cap * 8); s->idt.mask = cap - 1;
    }
    int* dup = reinterpret_cast<int*>(reinterpret_cast<char*>(s->idt.keys) + (size_t)cap * 12);
    HIPCHK(hipMemsetAsync(s->idt.keys, 0xFF, (size_t)cap * 8, s->stream));
    HIPCHK(hipMemsetAsync(dup, 0, 4, s->stream));
    track_launch_index_store(s->base, s->L.bytes, first, n, s->idt, dup, s->stream);
    HIPCHK(hipGetLastError());
    int h_dup = 0;
    HIPCHK(hipMemcpyAsync(&h_dup, dup, 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->idt_first = first; s->idt_n = n; s->idt_valid = !h_dup;
    if (h_dup) { corb_set_error("corb_mp_store_build_index: two of the slots hold the same map point id"); return CORB_ERR_ARG; }
    return CORB_OK;
}

extern "C" int corb_kf_store_count(CorbKfStore* s, int slot) { return (!s || slot < 0 || slot >= s->capacity) ? -1 : s->host[slot].n; }

namespace {
int check_stores(CorbKfStore* kf, int slot, CorbMpStore* mp, const CorbTrackCamera* cam, const char* who)
{
    if (!kf || !mp || !cam || slot < 0 || slot >= kf->capacity) { corb_set_error("%s: bad store / slot", who); return CORB_ERR_ARG; }
    if (kf->device != mp->device) { corb_set_error("%s: the stores live on different devices", who); return CORB_ERR_ARG; }
    if (kf->host[slot].n < 0) { corb_set_error("%s: slot %d is empty (or was filled without a host-known feature count)", who, slot); return CORB_ERR_ARG; }
    if (!mp->idt.keys || !mp->idt_valid) { corb_set_error("%s: the map-point store has no current id index (corb_mp_store_build_index after the last put / push)", who); return CORB_ERR_ARG; }
    if (cam->nlevels < 1 || cam->nlevels > CORB_MAX_LEVELS || !(cam->max_x > cam->min_x) || !(cam->max_y > cam->min_y)) { corb_set_error("%s: bad camera", who); return CORB_ERR_ARG; }
    return CORB_OK;
}
}  // namespace

extern "C" int corb_track_search_last_frame(CorbKfStore* frames, int cur_slot, int last_slot, CorbMpStore* map, const float* Tcw, const float* Tlw,
                                            const CorbTrackCamera* cam, float th, int mono, float nnratio, int check_orientation, int32_t* match, int* n_matches)
{
    int rc = check_stores(frames, cur_slot, map, cam, "corb_track_search_last_frame"); if (rc) return rc;
    if (!Tcw || !Tlw || !n_matches || last_slot < 0 || last_slot >= frames->capacity || last_slot == cur_slot || frames->host[last_slot].n < 0) {
        corb_set_error("corb_track_search_last_frame: bad argument"); return CORB_ERR_ARG;
    }
    const int n = frames->host[cur_slot].n, nq = frames->host[last_slot].n;
    *n_matches = 0;
    if (match) for (int i = 0; i < n; i++) match[i] = -1;
    if (n == 0 || nq == 0) return CORB_OK;
    if (n > 6000 || nq > 60000) { corb_set_error("corb_track_search_last_frame: frame too large"); return CORB_ERR_ARG; }
    rc = corb_select_device(frames->device); if (rc) return rc;
    // the two stores' own streams may still be filling the records: this call runs on the lane's stream after them
    std::lock_guard<std::mutex> lk(frames->mu); std::lock_guard<std::mutex> lk2(map->mu);       // (always in this order)
    HIPCHK(hipStreamSynchronize(frames->stream)); HIPCHK(hipStreamSynchronize(map->stream));
    CorbScratch pool(0);
    const RecLayout L(frames->F);
    char* cur = frames->rec(cur_slot); const char* last = frames->rec(last_slot);
    CorbLastPoint* lastp; unsigned long long* qdesc; unsigned char* claimed; CorbProjQuery* query; int *feat_cell, *cell_off, *cell_idx, *cand_cnt, *ev_feat, *ev_bin, *dmatch, *nm;
    unsigned long long* cand_key; unsigned char* cand_oct;
    HIPCHK(pool.alloc(&lastp, (size_t)nq)); HIPCHK(pool.alloc(&qdesc, (size_t)nq * 4)); HIPCHK(pool.alloc(&claimed, (size_t)n)); HIPCHK(pool.alloc(&query, (size_t)nq));
    HIPCHK(pool.alloc(&feat_cell, (size_t)n)); HIPCHK(pool.alloc(&cell_off, (size_t)PROJ_CELLS + 1)); HIPCHK(pool.alloc(&cell_idx, (size_t)n));
    HIPCHK(pool.alloc(&cand_key, (size_t)nq * PROJ_CAND_CAP)); HIPCHK(pool.alloc(&cand_oct, (size_t)nq * PROJ_CAND_CAP)); HIPCHK(pool.alloc(&cand_cnt, (size_t)nq));
    HIPCHK(pool.alloc(&ev_feat, (size_t)nq)); HIPCHK(pool.alloc(&ev_bin, (size_t)nq)); HIPCHK(pool.alloc(&nm, (size_t)n + 64)); dmatch = nm + 64;      // counts | matches as ONE block: one copy to the host
    HIPCHK(hipMemsetAsync(nm, 0, 8, pool.stream));
    TrackDev t; memset(&t, 0, sizeof(t));
    t.cur = cur; t.last = last; t.F = frames->F; t.mp_base = map->base; t.mp_bytes = map->L.bytes; t.idt = map->idt;
    t.lastp = lastp; t.qdesc = qdesc; t.claimed = claimed; t.match = dmatch; t.n_cur = n; t.n_last = nq;
    track_launch_prepare_last(t, pool.stream);
    CorbProjDev d; memset(&d, 0, sizeof(d));
    d.n = n; d.nq = nq; d.min_x = cam->min_x; d.min_y = cam->min_y; d.max_x = cam->max_x; d.max_y = cam->max_y;
    d.winv = (float)PROJ_COLS / (cam->max_x - cam->min_x); d.hinv = (float)PROJ_ROWS / (cam->max_y - cam->min_y);
    for (int l = 0; l < cam->nlevels; l++) d.scale[l] = cam->scale[l];
    d.nnratio = nnratio; d.ratio_test = 0; d.check_ori = check_orientation; d.check_uright = 1; d.th_dist = CORB_TH_HIGH;
    d.keys = reinterpret_cast<const CorbKeyPoint*>(cur + L.kp); d.u_right = reinterpret_cast<const float*>(cur + L.ur); d.desc = reinterpret_cast<const unsigned long long*>(cur + L.desc);
    d.claimed = claimed; d.qdesc = qdesc; d.query = query; d.feat_cell = feat_cell; d.cell_off = cell_off; d.cell_idx = cell_idx;
    d.cand_key = cand_key; d.cand_oct = cand_oct; d.cand_cnt = cand_cnt; d.ev_feat = ev_feat; d.ev_bin = ev_bin; d.match = dmatch; d.n_matches = nm; d.status = nm + 1;
    // the forward / backward test of the reference (ORBmatcher.cc:1480-1491): tlc = Rlw * twc + tlw, z against the baseline
    CorbProjPose pose; memcpy(pose.Tcw, Tcw, sizeof(float) * 16);
    pose.fx = cam->fx; pose.fy = cam->fy; pose.cx = cam->cx; pose.cy = cam->cy; pose.bf = cam->bf;
    {
        float twc[3], tlc2;
        for (int i = 0; i < 3; i++) twc[i] = -(Tcw[0 * 4 + i] * Tcw[3] + Tcw[1 * 4 + i] * Tcw[7] + Tcw[2 * 4 + i] * Tcw[11]);
        tlc2 = Tlw[8] * twc[0] + Tlw[9] * twc[1] + Tlw[10] * twc[2] + Tlw[11];
        pose.forward = (tlc2 > cam->mb && !mono) ? 1 : 0; pose.backward = (-tlc2 > cam->mb && !mono) ? 1 : 0;
    }
    corb_launch_projection(d, nullptr, lastp, &pose, th, pool.stream);
    track_launch_scatter_last(t, pool.stream);
    HIPCHK(hipGetLastError());
    static thread_local std::vector<int32_t> blk;            // (match is only handed over when the call succeeds)
    blk.resize((size_t)n + 64);
    HIPCHK(pool.d2h(blk.data(), nm, match ? ((size_t)n + 64) * 4 : 8));
    HIPCHK(pool.fetch_finish());
    const int* res = blk.data();
    if (res[1] != 0) { corb_set_error("corb_track_search_last_frame: more than %d candidates in one search window", PROJ_CAND_CAP); return CORB_ERR_OVERFLOW; }
    if (match) memcpy(match, blk.data() + 64, (size_t)n * 4);
    *n_matches = res[0];
    return CORB_OK;
}

extern "C" int corb_track_pose_optimization(CorbKfStore* frames, int slot, CorbMpStore* map, const CorbTrackCamera* cam, const float* Tcw_in, float* Tcw_out,
                                            int discard_outliers, uint8_t* outlier, int32_t* n_inliers)
{
    int rc = check_stores(frames, slot, map, cam, "corb_track_pose_optimization"); if (rc) return rc;
    if (!Tcw_in || !Tcw_out) { corb_set_error("corb_track_pose_optimization: bad argument"); return CORB_ERR_ARG; }
    const int n = frames->host[slot].n;
    memcpy(Tcw_out, Tcw_in, sizeof(float) * 16);
    if (n_inliers) *n_inliers = 0;
    if (outlier) memset(outlier, 0, (size_t)n);
    if (n == 0) return CORB_OK;
    rc = corb_select_device(frames->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(frames->mu); std::lock_guard<std::mutex> lk2(map->mu);       // (always in this order)
    HIPCHK(hipStreamSynchronize(frames->stream)); HIPCHK(hipStreamSynchronize(map->stream));
    CorbScratch pool(0);
    TrackPoseDev t; memset(&t, 0, sizeof(t));
    t.cur = frames->rec(slot); t.F = frames->F; t.n_cur = n; t.mp_base = map->base; t.mp_bytes = map->L.bytes; t.idt = map->idt;
    double *dpose, *dcam, *dlast; unsigned char* dact; int* dcnt;
    HIPCHK(pool.alloc(&t.edge_off, 2)); HIPCHK(pool.alloc(&t.stage_limit, 1)); HIPCHK(pool.alloc(&t.pt, (size_t)3 * n)); HIPCHK(pool.alloc(&t.obs, (size_t)3 * n)); HIPCHK(pool.alloc(&t.w, (size_t)n));
    HIPCHK(pool.alloc(&t.dim, (size_t)n)); HIPCHK(pool.alloc(&t.efeat, (size_t)n)); HIPCHK(pool.alloc(&dlast, (size_t)n)); HIPCHK(pool.alloc(&dact, (size_t)n)); HIPCHK(pool.alloc(&dcnt, 4));
    double h[12]; corb_pose_from_T(Tcw_in, h);
    h[7] = cam->fx; h[8] = cam->fy; h[9] = cam->cx; h[10] = cam->cy; h[11] = cam->bf;
    HIPCHK(pool.upload_block({{(void**)&dpose, h, sizeof(h)}}));
    dcam = dpose + 7;
    track_launch_pose_gather(t, pool.stream);
    CorbPoseDev d; memset(&d, 0, sizeof(d));
    d.n_problems = 1; d.n_stages = 4; corb_pose_optimization_stages(d.stages);
    d.edge_off = t.edge_off; d.pt = t.pt; d.obs = t.obs; d.w = t.w; d.dim = t.dim; d.cam = dcam; d.pose = dpose; d.last_chi2 = dlast; d.active = dact; d.counters = dcnt;
    d.stage_limit = t.stage_limit;                      // the edge count is on the device: the gather kernel turns it into the reference's early exits
    pose_launch_optimize(d, n, pool.stream);
    t.active = dact; t.pose = dpose; t.counters = dcnt; t.discard = discard_outliers ? 1 : 0;
    struct Res { double pose[7]; int cnt[4]; int E[2]; };           // (the finish kernel packs it: one copy instead of three)
    static_assert(sizeof(Res) == 80, "pose | counters | edge counts");
    double* dres; HIPCHK(pool.alloc(&dres, 10)); t.result = dres;
    track_launch_pose_finish(t, pool.stream);
    HIPCHK(hipGetLastError());
    Res* r = static_cast<Res*>(pool.pinned());
    HIPCHK(hipMemcpyAsync(r, dres, sizeof(Res), hipMemcpyDeviceToHost, pool.stream));
    std::vector<unsigned char> fl;
    if (outlier) { fl.resize((size_t)n); const RecLayout L(frames->F); HIPCHK(pool.d2h(fl.data(), t.cur + L.flags, (size_t)n)); }
    HIPCHK(pool.fetch_finish());
    if (r->cnt[2]) corb_pose_to_T(r->pose, Tcw_out);
    if (n_inliers) *n_inliers = r->E[1] < 3 ? 0 : r->cnt[3];          // `if(nInitialCorrespondences<3) return 0;`
    // (flags as they were BEFORE a discard would clear mvbOutlier: the caller sees which features the optimisation rejected)
    if (outlier) for (int i = 0; i < n; i++) outlier[i] = (fl[i] & (discard_outliers ? CORB_FEATURE_DISCARDED : CORB_FEATURE_OUTLIER)) ? 1 : 0;
    return CORB_OK;
}

extern "C" int corb_track_search_local_points(CorbKfStore* frames, int slot, CorbMpStore* map, const uint64_t* local_ids, int n_local, const CorbTrackCamera* cam,
                                              const float* Tcw, float log_scale_factor, float th, float nnratio, int32_t* match, CorbTrackedPoint* tracked,
                                              int* n_matches, int* n_in_view)
{
    int rc = check_stores(frames, slot, map, cam, "corb_track_search_local_points"); if (rc) return rc;
    if (!Tcw || !n_matches || n_local < 0 || (n_local > 0 && !local_ids) || !(log_scale_factor > 0)) { corb_set_error("corb_track_search_local_points: bad argument"); return CORB_ERR_ARG; }
    const int n = frames->host[slot].n, nq = n_local;
    *n_matches = 0; if (n_in_view) *n_in_view = 0;
    if (match) for (int i = 0; i < n; i++) match[i] = -1;
    if (tracked && nq) memset(tracked, 0, sizeof(CorbTrackedPoint) * (size_t)nq);
    if (n == 0) return CORB_OK;
    if (n > 6000 || nq > 60000) { corb_set_error("corb_track_search_local_points: too large (%d features, %d points)", n, nq); return CORB_ERR_ARG; }
    rc = corb_select_device(frames->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(frames->mu); std::lock_guard<std::mutex> lk2(map->mu);       // (always in this order)
    HIPCHK(hipStreamSynchronize(frames->stream)); HIPCHK(hipStreamSynchronize(map->stream));
    CorbScratch pool(0);
    const RecLayout L(frames->F);
    char* cur = frames->rec(slot);
    TrackLocalDev t; memset(&t, 0, sizeof(t));
    t.cur = cur; t.F = frames->F; t.n_cur = n; t.mp_base = map->base; t.mp_bytes = map->L.bytes; t.idt = map->idt; t.n_local = nq;
    unsigned int cap = 64; while (cap < 2u * (unsigned int)n) cap <<= 1;
    HIPCHK(pool.alloc(&t.inframe.keys, (size_t)cap)); HIPCHK(pool.alloc(&t.inframe.vals, (size_t)cap)); t.inframe.mask = cap - 1;
    HIPCHK(hipMemsetAsync(t.inframe.keys, 0xFF, (size_t)cap * 8, pool.stream));
    unsigned long long* dids = nullptr;
    const int nq1 = nq > 0 ? nq : 1;
    if (nq > 0) HIPCHK(pool.upload_block({{(void**)&dids, local_ids, (size_t)nq * 8}})); else HIPCHK(pool.alloc(&dids, 1));
    t.ids = dids;
    CorbProjQuery* query; int *feat_cell, *cell_off, *cell_idx, *cand_cnt, *ev_feat, *ev_bin, *dmatch, *nm; unsigned long long* cand_key; unsigned char* cand_oct;
    HIPCHK(pool.alloc(&t.tracked, (size_t)nq1)); HIPCHK(pool.alloc(&t.qdesc, (size_t)nq1 * 4)); HIPCHK(pool.alloc(&t.claimed, (size_t)n)); HIPCHK(pool.alloc(&query, (size_t)nq1));
    HIPCHK(pool.alloc(&feat_cell, (size_t)n)); HIPCHK(pool.alloc(&cell_off, (size_t)PROJ_CELLS + 1)); HIPCHK(pool.alloc(&cell_idx, (size_t)n));
    HIPCHK(pool.alloc(&cand_key, (size_t)nq1 * PROJ_CAND_CAP)); HIPCHK(pool.alloc(&cand_oct, (size_t)nq1 * PROJ_CAND_CAP)); HIPCHK(pool.alloc(&cand_cnt, (size_t)nq1));
    HIPCHK(pool.alloc(&ev_feat, (size_t)nq1)); HIPCHK(pool.alloc(&ev_bin, (size_t)nq1)); HIPCHK(pool.alloc(&nm, (size_t)n + 64)); dmatch = nm + 64;      // counts | matches as ONE block: one copy to the host
    HIPCHK(hipMemsetAsync(nm, 0, 16, pool.stream));
    t.match = dmatch; t.n_in_view = nm + 2;
    memcpy(t.Tcw, Tcw, sizeof(float) * 16);
    for (int i = 0; i < 3; i++)                          // mOw = -mRcw.t()*mtcw (Frame.cc UpdatePoseMatrices): a cv::gemm, double accumulation and one rounding
        t.Ow[i] = (float)(-((double)Tcw[0 * 4 + i] * (double)Tcw[3] + (double)Tcw[1 * 4 + i] * (double)Tcw[7] + (double)Tcw[2 * 4 + i] * (double)Tcw[11]));
    t.fx = cam->fx; t.fy = cam->fy; t.cx = cam->cx; t.cy = cam->cy; t.bf = cam->bf; t.min_x = cam->min_x; t.max_x = cam->max_x; t.min_y = cam->min_y; t.max_y = cam->max_y;
    t.log_scale = log_scale_factor; t.cos_limit = 0.5f; t.nlevels = cam->nlevels;
    track_launch_prepare_local(t, pool.stream);
    CorbProjDev d; memset(&d, 0, sizeof(d));
    d.n = n; d.nq = nq; d.min_x = cam->min_x; d.min_y = cam->min_y; d.max_x = cam->max_x; d.max_y = cam->max_y;
    d.winv = (float)PROJ_COLS / (cam->max_x - cam->min_x); d.hinv = (float)PROJ_ROWS / (cam->max_y - cam->min_y);
    for (int l = 0; l < cam->nlevels; l++) d.scale[l] = cam->scale[l];
    d.nnratio = nnratio; d.ratio_test = 1; d.check_ori = 0; d.check_uright = 1; d.th_dist = CORB_TH_HIGH;
    d.keys = reinterpret_cast<const CorbKeyPoint*>(cur + L.kp); d.u_right = reinterpret_cast<const float*>(cur + L.ur); d.desc = reinterpret_cast<const unsigned long long*>(cur + L.desc);
    d.claimed = t.claimed; d.qdesc = t.qdesc; d.query = query; d.feat_cell = feat_cell; d.cell_off = cell_off; d.cell_idx = cell_idx;
    d.cand_key = cand_key; d.cand_oct = cand_oct; d.cand_cnt = cand_cnt; d.ev_feat = ev_feat; d.ev_bin = ev_bin; d.match = dmatch; d.n_matches = nm; d.status = nm + 1;
    if (nq > 0) {
        corb_launch_projection(d, t.tracked, nullptr, nullptr, th, pool.stream);
        track_launch_scatter_local(t, pool.stream);
    }
    HIPCHK(hipGetLastError());
    static thread_local std::vector<int32_t> blk; std::vector<CorbTrackedPoint> tr2;
    blk.resize((size_t)n + 64);
    HIPCHK(pool.d2h(blk.data(), nm, (match && nq > 0) ? ((size_t)n + 64) * 4 : 16));
    if (tracked && nq > 0) { tr2.resize((size_t)nq); HIPCHK(pool.d2h(tr2.data(), t.tracked, sizeof(CorbTrackedPoint) * (size_t)nq)); }
    HIPCHK(pool.fetch_finish());
    const int* res = blk.data();
    if (res[1] != 0) { corb_set_error("corb_track_search_local_points: more than %d candidates in one search window", PROJ_CAND_CAP); return CORB_ERR_OVERFLOW; }
    if (match && nq > 0) memcpy(match, blk.data() + 64, (size_t)n * 4);
    if (tracked && nq > 0) memcpy(tracked, tr2.data(), sizeof(CorbTrackedPoint) * (size_t)nq);
    *n_matches = res[0]; if (n_in_view) *n_in_view = res[2];
    return CORB_OK;
}

// ---- int ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th) on records (see include/corb_accel.h) ----
extern "C" int corb_fuse_store(CorbKfStore* kf, int slot, CorbMpStore* map, const int32_t* mp_slots, int n_points, const CorbTrackCamera* cam,
                               const float* Tcw, float log_scale_factor, float th, int apply, int32_t* best_idx, int32_t* best_dist, uint8_t* action, int* n_fused)
{
    if (!kf || !map || !cam || slot < 0 || slot >= kf->capacity || n_points < 0 || (n_points > 0 && (!mp_slots || !best_idx || !best_dist)) || !Tcw || !n_fused || !(log_scale_factor > 0)) {
        corb_set_error("corb_fuse_store: bad argument"); return CORB_ERR_ARG;
    }
    if (kf->device != map->device) { corb_set_error("corb_fuse_store: the stores live on different devices"); return CORB_ERR_ARG; }
    if (kf->host[slot].n < 0) { corb_set_error("corb_fuse_store: slot %d is empty (or was filled without a host-known feature count)", slot); return CORB_ERR_ARG; }
    if (cam->nlevels < 1 || cam->nlevels > CORB_MAX_LEVELS || !(cam->max_x > cam->min_x) || !(cam->max_y > cam->min_y)) { corb_set_error("corb_fuse_store: bad camera"); return CORB_ERR_ARG; }
    for (int i = 0; i < n_points; i++) if (mp_slots[i] < 0 || mp_slots[i] >= map->capacity) { corb_set_error("corb_fuse_store: map-point slot out of range"); return CORB_ERR_ARG; }
    const int n = kf->host[slot].n, nq = n_points;
    *n_fused = 0;
    for (int i = 0; i < nq; i++) { best_idx[i] = -1; best_dist[i] = 256; if (action) action[i] = 0; }
    if (n == 0 || nq == 0) return CORB_OK;
    if (n > 6000 || nq > 60000) { corb_set_error("corb_fuse_store: too large (%d features, %d points)", n, nq); return CORB_ERR_ARG; }
    int rc = corb_select_device(kf->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(kf->mu); std::lock_guard<std::mutex> lk2(map->mu);       // (always in this order)
    HIPCHK(hipStreamSynchronize(kf->stream)); HIPCHK(hipStreamSynchronize(map->stream));
    CorbScratch pool(0);
    const RecLayout L(kf->F);
    char* rec = kf->rec(slot);
    FuseStoreDev t; memset(&t, 0, sizeof(t));
    t.kf_rec = rec; t.F = kf->F; t.n_feat = n; t.mp_base = map->base; t.mp_bytes = map->L.bytes; t.max_obs = map->O; t.n_points = nq; t.apply = apply ? 1 : 0;
    int* dslots = nullptr;
    HIPCHK(pool.upload_block({{(void**)&dslots, mp_slots, (size_t)nq * 4}}));
    t.mp_slots = dslots;
    CorbProjQuery* query; int *feat_cell, *cell_off, *cell_idx, *cand_cnt, *ev_feat, *ev_bin, *dmatch, *nm, *bi, *bd; unsigned long long* cand_key; unsigned char *cand_oct, *claimed;
    HIPCHK(pool.alloc(&t.pts, (size_t)nq)); HIPCHK(pool.alloc(&t.qdesc, (size_t)nq * 4)); HIPCHK(pool.alloc(&t.claim, (size_t)n)); HIPCHK(pool.alloc(&t.action, (size_t)nq));
    HIPCHK(pool.alloc(&claimed, (size_t)n)); HIPCHK(pool.alloc(&query, (size_t)nq));
    HIPCHK(pool.alloc(&feat_cell, (size_t)n)); HIPCHK(pool.alloc(&cell_off, (size_t)PROJ_CELLS + 1)); HIPCHK(pool.alloc(&cell_idx, (size_t)n));
    HIPCHK(pool.alloc(&cand_key, 1)); HIPCHK(pool.alloc(&cand_oct, 8)); HIPCHK(pool.alloc(&cand_cnt, (size_t)nq));
    HIPCHK(pool.alloc(&ev_feat, (size_t)nq)); HIPCHK(pool.alloc(&ev_bin, (size_t)nq)); HIPCHK(pool.alloc(&dmatch, (size_t)n)); HIPCHK(pool.alloc(&nm, 2));
    HIPCHK(pool.alloc(&bi, (size_t)nq)); HIPCHK(pool.alloc(&bd, (size_t)nq));
    HIPCHK(hipMemsetAsync(nm, 0, 8, pool.stream)); HIPCHK(hipMemsetAsync(claimed, 0, (size_t)n, pool.stream));
    t.best_idx = bi;
    fuse_launch_prepare(t, pool.stream);
    // the search itself: the kernels of corb_fuse on the record's arrays (Tcw: pKF->GetPose(); Ow = pKF->GetCameraCenter() = -Rcw^T tcw, KeyFrame.cc:120-135)
    CorbProjTf tf; memset(&tf, 0, sizeof(tf));
    tf.fx = cam->fx; tf.fy = cam->fy; tf.cx = cam->cx; tf.cy = cam->cy; tf.bf = cam->bf; tf.log_scale = log_scale_factor; tf.th = th; tf.nlevels = cam->nlevels;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) tf.A[i * 4 + j] = Tcw[i * 4 + j];
    for (int i = 0; i < 3; i++) { double sum = 0; for (int k = 0; k < 3; k++) sum += (double)(-Tcw[k * 4 + i]) * (double)Tcw[k * 4 + 3]; tf.Ow[i] = (float)sum; }
    tf.check_normal = 1; tf.lvl_hi = 0; tf.invz_double = 0;
    CorbProjDev d; memset(&d, 0, sizeof(d));
    d.n = n; d.nq = nq; d.min_x = cam->min_x; d.min_y = cam->min_y; d.max_x = cam->max_x; d.max_y = cam->max_y;
    d.winv = (float)PROJ_COLS / (cam->max_x - cam->min_x); d.hinv = (float)PROJ_ROWS / (cam->max_y - cam->min_y);
    for (int l = 0; l < cam->nlevels; l++) { d.scale[l] = cam->scale[l]; const float s2 = cam->scale[l] * cam->scale[l]; d.inv_sigma2[l] = 1.0f / s2; }    // mvLevelSigma2 / mvInvLevelSigma2 (ORBextractor.cc:418-430)
    d.nnratio = 0.f; d.ratio_test = 0; d.check_ori = 0; d.check_uright = 0; d.th_dist = CORB_TH_LOW; d.chi2_check = 1;
    d.keys = reinterpret_cast<const CorbKeyPoint*>(rec + L.kp); d.u_right = reinterpret_cast<const float*>(rec + L.ur); d.desc = reinterpret_cast<const unsigned long long*>(rec + L.desc);
    d.claimed = claimed; d.qdesc = t.qdesc; d.query = query; d.feat_cell = feat_cell; d.cell_off = cell_off; d.cell_idx = cell_idx;
    d.cand_key = cand_key; d.cand_oct = cand_oct; d.cand_cnt = cand_cnt; d.ev_feat = ev_feat; d.ev_bin = ev_bin; d.match = dmatch; d.n_matches = nm; d.status = nm + 1;
    d.best_idx = bi; d.best_dist = bd;
    corb_launch_projection_points(d, t.pts, tf, 0, pool.stream);
    fuse_launch_apply(t, pool.stream);
    HIPCHK(hipGetLastError());
    static thread_local std::vector<int32_t> h_bi, h_bd; static thread_local std::vector<uint8_t> h_act;
    h_bi.resize((size_t)nq); h_bd.resize((size_t)nq); h_act.resize((size_t)nq);
    HIPCHK(pool.d2h(h_bi.data(), bi, (size_t)nq * 4)); HIPCHK(pool.d2h(h_bd.data(), bd, (size_t)nq * 4)); HIPCHK(pool.d2h(h_act.data(), t.action, (size_t)nq));
    HIPCHK(pool.fetch_finish());
    int nf = 0, full = 0;
    for (int i = 0; i < nq; i++) { best_idx[i] = h_bi[i]; best_dist[i] = h_bd[i]; if (action) action[i] = h_act[i]; nf += h_bi[i] >= 0; full += h_act[i] == 3; }
    *n_fused = nf;
    if (full) { corb_set_error("corb_fuse_store: %d map points have no room for another observation (max_observations = %d); they were not added", full, map->O); return CORB_ERR_CAPACITY; }
    return CORB_OK;
}

namespace {
// the matcher's scratch for `n` target features and `nq` queries (greedy: candidate lists per query)
struct ProjScratch { CorbProjQuery* query; int *feat_cell, *cell_off, *cell_idx, *cand_cnt, *ev_feat, *ev_bin, *dmatch, *nm, *bi, *bd; unsigned long long* cand_key; unsigned char* cand_oct; };
int proj_scratch(CorbScratch& pool, int n, int nq, bool greedy, ProjScratch& p)
{
    const size_t nq1 = nq > 0 ? nq : 1, n1 = n > 0 ? n : 1;
    HIPCHK(pool.alloc(&p.query, nq1)); HIPCHK(pool.alloc(&p.feat_cell, n1)); HIPCHK(pool.alloc(&p.cell_off, (size_t)PROJ_CELLS + 1)); HIPCHK(pool.alloc(&p.cell_idx, n1));
    HIPCHK(pool.alloc(&p.cand_key, greedy ? nq1 * PROJ_CAND_CAP : 1)); HIPCHK(pool.alloc(&p.cand_oct, greedy ? nq1 * PROJ_CAND_CAP : 8)); HIPCHK(pool.alloc(&p.cand_cnt, nq1));
    HIPCHK(pool.alloc(&p.ev_feat, nq1)); HIPCHK(pool.alloc(&p.ev_bin, nq1)); HIPCHK(pool.alloc(&p.dmatch, n1)); HIPCHK(pool.alloc(&p.nm, 2));
    HIPCHK(pool.alloc(&p.bi, nq1)); HIPCHK(pool.alloc(&p.bd, nq1));
    HIPCHK(hipMemsetAsync(p.nm, 0, 8, pool.stream));
    return CORB_OK;
}
void proj_dev(CorbProjDev& d, const CorbTrackCamera* cam, const char* rec, const RecLayout& L, int n, int nq, const ProjScratch& p)
{
    memset(&d, 0, sizeof(d));
    d.n = n; d.nq = nq; d.min_x = cam->min_x; d.min_y = cam->min_y; d.max_x = cam->max_x; d.max_y = cam->max_y;
    d.winv = (float)PROJ_COLS / (cam->max_x - cam->min_x); d.hinv = (float)PROJ_ROWS / (cam->max_y - cam->min_y);
    for (int l = 0; l < cam->nlevels; l++) { d.scale[l] = cam->scale[l]; d.inv_sigma2[l] = 1.0f / (cam->scale[l] * cam->scale[l]); }
    d.keys = reinterpret_cast<const CorbKeyPoint*>(rec + L.kp); d.u_right = reinterpret_cast<const float*>(rec + L.ur); d.desc = reinterpret_cast<const unsigned long long*>(rec + L.desc);
    d.query = p.query; d.feat_cell = p.feat_cell; d.cell_off = p.cell_off; d.cell_idx = p.cell_idx; d.cand_key = p.cand_key; d.cand_oct = p.cand_oct; d.cand_cnt = p.cand_cnt;
    d.ev_feat = p.ev_feat; d.ev_bin = p.ev_bin; d.match = p.dmatch; d.n_matches = p.nm; d.status = p.nm + 1; d.best_idx = p.bi; d.best_dist = p.bd;
}
void tf_from_camera(CorbProjTf& tf, const CorbTrackCamera* cam, float log_scale_factor, float th)
{
    memset(&tf, 0, sizeof(tf));
    tf.fx = cam->fx; tf.fy = cam->fy; tf.cx = cam->cx; tf.cy = cam->cy; tf.bf = cam->bf; tf.log_scale = log_scale_factor; tf.th = th; tf.nlevels = cam->nlevels;
}
}  // namespace

// ---- int ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>& sAlreadyFound, th, ORBdist) on records (see include/corb_accel.h) ----
extern "C" int corb_track_search_reloc(CorbKfStore* frames, int cur_slot, CorbKfStore* kfs, int kf_slot, CorbMpStore* map, const CorbTrackCamera* cam,
                                       const float* Tcw, float log_scale_factor, float th, int orb_dist, int check_orientation, int32_t* match, int* n_matches)
{
    int rc = check_stores(frames, cur_slot, map, cam, "corb_track_search_reloc"); if (rc) return rc;
    if (!kfs || kf_slot < 0 || kf_slot >= kfs->capacity || kfs->host[kf_slot].n < 0 || kfs->device != frames->device || !Tcw || !n_matches || !(log_scale_factor > 0) ||
        (kfs == frames && kf_slot == cur_slot)) { corb_set_error("corb_track_search_reloc: bad argument"); return CORB_ERR_ARG; }
    const int n = frames->host[cur_slot].n, nq = kfs->host[kf_slot].n;
    *n_matches = 0;
    if (match) for (int i = 0; i < n; i++) match[i] = -1;
    if (n == 0 || nq == 0) return CORB_OK;
    if (n > 6000 || nq > 60000) { corb_set_error("corb_track_search_reloc: too large (%d features, %d points)", n, nq); return CORB_ERR_ARG; }
    rc = corb_select_device(frames->device); if (rc) return rc;
    // lock order: keyframe stores (by address when there are two), then the map
    std::unique_lock<std::mutex> lk_a, lk_b;
    if (kfs == frames) lk_a = std::unique_lock<std::mutex>(frames->mu);
    else { CorbKfStore* lo = frames < kfs ? frames : kfs; CorbKfStore* hi = frames < kfs ? kfs : frames; lk_a = std::unique_lock<std::mutex>(lo->mu); lk_b = std::unique_lock<std::mutex>(hi->mu); }
    std::lock_guard<std::mutex> lk2(map->mu);
    HIPCHK(hipStreamSynchronize(frames->stream)); if (kfs != frames) HIPCHK(hipStreamSynchronize(kfs->stream)); HIPCHK(hipStreamSynchronize(map->stream));
    CorbScratch pool(0);
    RelocStoreDev t; memset(&t, 0, sizeof(t));
    t.cur = frames->rec(cur_slot); t.F_cur = frames->F; t.n_cur = n; t.kf = kfs->rec(kf_slot); t.F_kf = kfs->F; t.n_kf = nq;
    t.mp_base = map->base; t.mp_bytes = map->L.bytes; t.idt = map->idt;
    unsigned int cap = 64; while (cap < 2u * (unsigned int)n) cap <<= 1;
    HIPCHK(pool.alloc(&t.inframe.keys, (size_t)cap)); HIPCHK(pool.alloc(&t.inframe.vals, (size_t)cap)); t.inframe.mask = cap - 1;
    HIPCHK(hipMemsetAsync(t.inframe.keys, 0xFF, (size_t)cap * 8, pool.stream));
    HIPCHK(pool.alloc(&t.pts, (size_t)nq)); HIPCHK(pool.alloc(&t.qdesc, (size_t)nq * 4)); HIPCHK(pool.alloc(&t.claimed, (size_t)n));
    ProjScratch ps; rc = proj_scratch(pool, n, nq, true, ps); if (rc) return rc;
    t.match = ps.dmatch;
    reloc_launch_prepare(t, pool.stream);
    // the transform of corb_search_by_projection_reloc: Rcw / tcw of the frame, Ow = -Rcw^T tcw, closed image test, invz in double, octaves [level - 1, level + 1]
    CorbProjTf tf; tf_from_camera(tf, cam, log_scale_factor, th);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) tf.A[i * 4 + j] = Tcw[i * 4 + j];
    for (int i = 0; i < 3; i++) { double sum = 0; for (int k = 0; k < 3; k++) sum += (double)(-Tcw[k * 4 + i]) * (double)Tcw[k * 4 + 3]; tf.Ow[i] = (float)sum; }
    tf.reloc = 1; tf.invz_double = 1; tf.lvl_hi = 1;
    const RecLayout L(frames->F);
    CorbProjDev d; proj_dev(d, cam, t.cur, L, n, nq, ps);
    d.claimed = t.claimed; d.qdesc = t.qdesc;
    d.nnratio = 0.f; d.ratio_test = 0; d.check_ori = check_orientation ? 1 : 0; d.check_uright = 0; d.th_dist = orb_dist; d.chi2_check = 0;
    corb_launch_projection_points(d, t.pts, tf, 1, pool.stream);
    reloc_launch_scatter(t, pool.stream);
    HIPCHK(hipGetLastError());
    int* res = static_cast<int*>(pool.pinned());
    HIPCHK(hipMemcpyAsync(res, ps.nm, 8, hipMemcpyDeviceToHost, pool.stream));
    std::vector<int32_t> m2;
    if (match) { m2.resize((size_t)n); HIPCHK(pool.d2h(m2.data(), ps.dmatch, (size_t)n * 4)); }
    HIPCHK(pool.fetch_finish());
    if (res[1] != 0) { corb_set_error("corb_track_search_reloc: more than %d candidates in one search window", PROJ_CAND_CAP); return CORB_ERR_OVERFLOW; }
    if (match) memcpy(match, m2.data(), (size_t)n * 4);
    *n_matches = res[0];
    return CORB_OK;
}

// ---- int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) on records (see include/corb_accel.h) ----
extern "C" int corb_search_by_projection_scw_store(CorbKfStore* kf, int slot, CorbMpStore* map, const int32_t* mp_slots, int n_points, const CorbTrackCamera* cam,
                                                   const float* Scw, float log_scale_factor, float th, uint64_t* matched_ids, int32_t* match, int* n_matches)
{
    if (!kf || !map || !cam || slot < 0 || slot >= kf->capacity || kf->device != map->device || kf->host[slot].n < 0) { corb_set_error("corb_search_by_projection_scw_store: bad store / slot"); return CORB_ERR_ARG; }
    if (cam->nlevels < 1 || cam->nlevels > CORB_MAX_LEVELS || !(cam->max_x > cam->min_x) || !(cam->max_y > cam->min_y)) { corb_set_error("corb_search_by_projection_scw_store: bad camera"); return CORB_ERR_ARG; }
    int rc = CORB_OK;
    if (n_points < 0 || (n_points > 0 && !mp_slots) || !Scw || !n_matches || !(log_scale_factor > 0) || (kf->host[slot].n > 0 && !matched_ids)) {
        corb_set_error("corb_search_by_projection_scw_store: bad argument"); return CORB_ERR_ARG;
    }
    for (int i = 0; i < n_points; i++) if (mp_slots[i] < 0 || mp_slots[i] >= map->capacity) { corb_set_error("corb_search_by_projection_scw_store: map-point slot out of range"); return CORB_ERR_ARG; }
    const int n = kf->host[slot].n, nq = n_points;
    *n_matches = 0;
    if (match) for (int i = 0; i < n; i++) match[i] = -1;
    if (n == 0 || nq == 0) return CORB_OK;
    if (n > 6000 || nq > 60000) { corb_set_error("corb_search_by_projection_scw_store: too large (%d features, %d points)", n, nq); return CORB_ERR_ARG; }
    rc = corb_select_device(kf->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(kf->mu); std::lock_guard<std::mutex> lk2(map->mu);       // (always in this order)
    HIPCHK(hipStreamSynchronize(kf->stream)); HIPCHK(hipStreamSynchronize(map->stream));
    CorbScratch pool(0);
    ScwStoreDev t; memset(&t, 0, sizeof(t));
    t.kf_rec = kf->rec(slot); t.F = kf->F; t.n_feat = n; t.mp_base = map->base; t.mp_bytes = map->L.bytes; t.n_points = nq;
    int* dslots = nullptr; unsigned long long* dmatched = nullptr;
    HIPCHK(pool.upload_block({{(void**)&dslots, mp_slots, (size_t)nq * 4}, {(void**)&dmatched, matched_ids, (size_t)n * 8}}));
    t.mp_slots = dslots; t.matched = dmatched;
    unsigned int cap = 64; while (cap < 2u * (unsigned int)n) cap <<= 1;
    HIPCHK(pool.alloc(&t.found.keys, (size_t)cap)); HIPCHK(pool.alloc(&t.found.vals, (size_t)cap)); t.found.mask = cap - 1;
    HIPCHK(hipMemsetAsync(t.found.keys, 0xFF, (size_t)cap * 8, pool.stream));
    HIPCHK(pool.alloc(&t.pts, (size_t)nq)); HIPCHK(pool.alloc(&t.qdesc, (size_t)nq * 4)); HIPCHK(pool.alloc(&t.claimed, (size_t)n));
    ProjScratch ps; rc = proj_scratch(pool, n, nq, true, ps); if (rc) return rc;
    t.match = ps.dmatch;
    scw_launch_prepare(t, pool.stream);
    // the transform of corb_search_by_projection_scw: Scw decomposed (:434-438), Fuse's gates, a float 1/z, octaves [level - 1, level], TH_LOW, no chi2 test
    CorbProjTf tf; tf_from_camera(tf, cam, log_scale_factor, th);
    {
        const double dd = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
        const float scw = (float)std::sqrt(dd);
        const float inv = (float)(1.0 / (double)scw);
        float M[12];
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[i * 4 + j] = Scw[i * 4 + j] * inv; M[i * 4 + 3] = Scw[i * 4 + 3] * inv; }
        for (int i = 0; i < 12; i++) tf.A[i] = M[i];
        for (int i = 0; i < 3; i++) { double sum = 0; for (int k = 0; k < 3; k++) sum += (double)(-M[k * 4 + i]) * (double)M[k * 4 + 3]; tf.Ow[i] = (float)sum; }
    }
    tf.invz_double = 0; tf.check_normal = 1; tf.lvl_hi = 0;
    const RecLayout L(kf->F);
    CorbProjDev d; proj_dev(d, cam, t.kf_rec, L, n, nq, ps);
    d.claimed = t.claimed; d.qdesc = t.qdesc;
    d.nnratio = 0.f; d.ratio_test = 0; d.check_ori = 0; d.check_uright = 0; d.th_dist = CORB_TH_LOW; d.chi2_check = 0;
    corb_launch_projection_points(d, t.pts, tf, 1, pool.stream);
    scw_launch_scatter(t, pool.stream);
    HIPCHK(hipGetLastError());
    int* res = static_cast<int*>(pool.pinned());
    HIPCHK(hipMemcpyAsync(res, ps.nm, 8, hipMemcpyDeviceToHost, pool.stream));
    std::vector<int32_t> m2; std::vector<uint64_t> ids2((size_t)n);
    if (match) { m2.resize((size_t)n); HIPCHK(pool.d2h(m2.data(), ps.dmatch, (size_t)n * 4)); }
    HIPCHK(pool.d2h(ids2.data(), dmatched, (size_t)n * 8));
    HIPCHK(pool.fetch_finish());
    if (res[1] != 0) { corb_set_error("corb_search_by_projection_scw_store: more than %d candidates in one search window", PROJ_CAND_CAP); return CORB_ERR_OVERFLOW; }
    if (match) memcpy(match, m2.data(), (size_t)n * 4);
    memcpy(matched_ids, ids2.data(), (size_t)n * 8);
    *n_matches = res[0];
    return CORB_OK;
}

// ---- int ORBmatcher::SearchBySim3(KeyFrame*, KeyFrame*, vector<MapPoint*>& vpMatches12, s12, R12, t12, th) on records (see include/corb_accel.h) ----
extern "C" int corb_search_by_sim3_store(CorbKfStore* kf, int slot1, int slot2, CorbMpStore* map, const CorbTrackCamera* cam, float log_scale_factor,
                                         const float* T1w, const float* T2w, const uint64_t* matched12_ids, float s12, const float* R12, const float* t12, float th,
                                         int32_t* match12, uint64_t* match12_ids, int* n_found)
{
    int rc = check_stores(kf, slot1, map, cam, "corb_search_by_sim3_store"); if (rc) return rc;
    if (slot2 < 0 || slot2 >= kf->capacity || slot2 == slot1 || kf->host[slot2].n < 0 || !T1w || !T2w || !R12 || !t12 || !match12 || !n_found || !(log_scale_factor > 0) || !(s12 > 0)) {
        corb_set_error("corb_search_by_sim3_store: bad argument"); return CORB_ERR_ARG;
    }
    const int N1 = kf->host[slot1].n, N2 = kf->host[slot2].n;
    *n_found = 0;
    for (int i = 0; i < N1; i++) { match12[i] = -1; if (match12_ids) match12_ids[i] = CORB_NO_MAP_POINT; }
    if (N1 == 0 || N2 == 0) return CORB_OK;
    if (N1 > 6000 || N2 > 6000) { corb_set_error("corb_search_by_sim3_store: keyframe too large"); return CORB_ERR_ARG; }
    rc = corb_select_device(kf->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(kf->mu); std::lock_guard<std::mutex> lk2(map->mu);
    HIPCHK(hipStreamSynchronize(kf->stream)); HIPCHK(hipStreamSynchronize(map->stream));
    CorbScratch pool(0);
    Sim3StoreDev t; memset(&t, 0, sizeof(t));
    t.kf1 = kf->rec(slot1); t.kf2 = kf->rec(slot2); t.F = kf->F; t.n1 = N1; t.n2 = N2;
    t.mp_base = map->base; t.mp_bytes = map->L.bytes; t.max_obs = map->O; t.idt = map->idt;
    unsigned long long* dm = nullptr;
    if (matched12_ids) { HIPCHK(pool.upload_block({{(void**)&dm, matched12_ids, (size_t)N1 * 8}})); t.matched12 = dm; }
    HIPCHK(pool.alloc(&t.already1, (size_t)N1)); HIPCHK(pool.alloc(&t.already2, (size_t)N2));
    HIPCHK(hipMemsetAsync(t.already2, 0, (size_t)N2, pool.stream));
    HIPCHK(pool.alloc(&t.pts1, (size_t)N1)); HIPCHK(pool.alloc(&t.pts2, (size_t)N2)); HIPCHK(pool.alloc(&t.qdesc1, (size_t)N1 * 4)); HIPCHK(pool.alloc(&t.qdesc2, (size_t)N2 * 4));
    sim3_launch_prepare(t, pool.stream);
    // sR12 = s12*R12 ; sR21 = (1.0/s12)*R12.t() ; t21 = -sR21*t12   (:1262-1264), as corb_search_by_sim3
    float sR12[9], sR21[9], t21[3];
    const float is = (float)(1.0 / (double)s12);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { sR12[i * 3 + j] = R12[i * 3 + j] * s12; sR21[i * 3 + j] = R12[j * 3 + i] * is; }
    for (int i = 0; i < 3; i++) { double sum = 0; for (int k = 0; k < 3; k++) sum += (double)(-sR21[i * 3 + k]) * (double)t12[k]; t21[i] = (float)sum; }
    const RecLayout L(kf->F);
    ProjScratch pa, pb;
    rc = proj_scratch(pool, N2, N1, false, pa); if (rc) return rc;            // KF1's points into KF2
    rc = proj_scratch(pool, N1, N2, false, pb); if (rc) return rc;            // KF2's points into KF1
    unsigned char* zero_claimed; HIPCHK(pool.alloc(&zero_claimed, (size_t)(N1 > N2 ? N1 : N2)));
    HIPCHK(hipMemsetAsync(zero_claimed, 0, (size_t)(N1 > N2 ? N1 : N2), pool.stream));
    auto direction = [&](const char* recB, int nB, const float* TAw, const float* sR, const float* tt, const CorbMapPointView* pts, const unsigned long long* qd, int nq, const ProjScratch& ps) {
        CorbProjTf tf; tf_from_camera(tf, cam, log_scale_factor, th);             // the intrinsics of both directions are pKF1's (:1247-1250)
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) tf.A[i * 4 + j] = TAw[i * 4 + j];
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) tf.B[i * 4 + j] = sR[i * 3 + j]; tf.B[i * 4 + 3] = tt[i]; }
        tf.two = 1; tf.invz_double = 1; tf.dist_from_cam = 1; tf.lvl_hi = 0;
        CorbProjDev d; proj_dev(d, cam, recB, L, nB, nq, ps);
        d.claimed = zero_claimed; d.qdesc = qd;
        d.nnratio = 0.f; d.ratio_test = 0; d.check_ori = 0; d.check_uright = 0; d.th_dist = CORB_TH_HIGH; d.chi2_check = 0;
        corb_launch_projection_points(d, pts, tf, 0, pool.stream);
    };
    direction(t.kf2, N2, T1w, sR21, t21, t.pts1, t.qdesc1, N1, pa);
    direction(t.kf1, N1, T2w, sR12, t12, t.pts2, t.qdesc2, N2, pb);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> m1((size_t)N1), m2((size_t)N2); std::vector<unsigned long long> ids2;
    HIPCHK(pool.d2h(m1.data(), pa.bi, (size_t)N1 * 4)); HIPCHK(pool.d2h(m2.data(), pb.bi, (size_t)N2 * 4));
    if (match12_ids) { ids2.resize((size_t)N2); HIPCHK(pool.d2h(ids2.data(), t.kf2 + L.mp_id, (size_t)N2 * 8)); }
    HIPCHK(pool.fetch_finish());
    int nf = 0;                                                               // check agreement (:1452-1465)
    for (int i1 = 0; i1 < N1; i1++) {
        const int idx2 = m1[i1];
        if (idx2 >= 0 && idx2 < N2 && m2[idx2] == i1) { match12[i1] = idx2; if (match12_ids) match12_ids[i1] = ids2[idx2]; nf++; }
    }
    *n_found = nf;
    return CORB_OK;
}
