// corb_store.cpp -- device-resident keyframe store, matchers on store slots, and the RCCL map push (see include/corb_accel.h).
// Replaces, for the hot path, what the reference moves as boost text archives through ROS services: the per-keyframe payload of
// corbslam_client/include/KeyFrame.h:59-87 (keypoints, descriptors, mvuRight, mvDepth, FeatureVector) and the client -> server batch of
// corbslam_client/src/Cache.cc:322-375 / DataDriver.cc:135-193.  One fixed-size SoA record per keyframe in HBM; a push is ncclSend / ncclRecv of
// whole records between the ranks' device buffers (xGMI), no serialisation, no host staging.
#include "match_internal.h"
#include "store_internal.h"
#include "store_host.h"
#include "corb_workspace.h"
#include <dlfcn.h>
#include <string>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

extern "C" int corb_kf_store_create(int device, int capacity, int max_features, CorbKfStore** out)
{
    if (!out || capacity < 1 || max_features < 1 || max_features > 65535) { corb_set_error("corb_kf_store_create: bad argument"); return CORB_ERR_ARG; }
    *out = nullptr;
    int rc = corb_select_device(device); if (rc) return rc;
    CorbKfStore* s = new CorbKfStore();
    s->device = device; s->capacity = capacity; s->F = max_features; s->L = RecLayout(max_features); s->host.resize(capacity);
    if (hipMalloc((void**)&s->base, (size_t)capacity * s->L.bytes) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev, hipEventDisableTiming) != hipSuccess) {
        corb_set_error("corb_kf_store_create: %d keyframes x %zu bytes: allocation failed", capacity, s->L.bytes);
        if (s->base) (void)hipFree(s->base);
        delete s; return CORB_ERR_HIP;
    }
    (void)hipMemsetAsync(s->base, 0, (size_t)capacity * s->L.bytes, s->stream);
    (void)hipStreamSynchronize(s->stream);
    *out = s;
    return CORB_OK;
}
extern "C" void corb_kf_store_destroy(CorbKfStore* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->ev) (void)hipEventDestroy(s->ev);
    if (s->base) (void)hipFree(s->base);
    delete s;
}
extern "C" int corb_kf_store_record_bytes(const CorbKfStore* s) { return s ? (int)s->L.bytes : 0; }

static int slot_ok(CorbKfStore* s, int slot, const char* who)
{
    if (!s || slot < 0 || slot >= s->capacity) { corb_set_error("%s: bad store / slot", who); return CORB_ERR_ARG; }
    return CORB_OK;
}
// counts and node ids of a slot whose record changed on the device (filled from the front-end or received by a push)
static int refresh_host(CorbKfStore* s, int slot)
{
    CorbKfStore::Host& h = s->host[slot];
    if (h.header_valid) return CORB_OK;
    int hdr[4];
    HIPCHK(hipMemcpyAsync(hdr, s->rec(slot), sizeof(hdr), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    h.n = hdr[0]; h.n_nodes = hdr[1]; memcpy(&h.id, &hdr[2], 8);
    if (h.n < 0 || h.n > s->F || h.n_nodes < 0 || h.n_nodes > s->F) { corb_set_error("keyframe store: slot %d holds a corrupt record", slot); return CORB_ERR_ARG; }
    h.node_id.resize(h.n_nodes);
    if (h.n_nodes) {
        std::vector<int32_t> off((size_t)h.n_nodes + 1);
        HIPCHK(hipMemcpyAsync(h.node_id.data(), s->rec(slot) + s->L.fv_node, (size_t)h.n_nodes * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(off.data(), s->rec(slot) + s->L.fv_off, off.size() * 4, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        bool ok = off[0] == 0 && off[h.n_nodes] <= s->F;
        for (int k = 0; k < h.n_nodes && ok; k++) ok = off[k] >= 0 && off[k] <= off[k + 1];
        if (!ok) { h.n = -1; corb_set_error("keyframe store: slot %d holds a corrupt FeatureVector", slot); return CORB_ERR_ARG; }
    }
    h.header_valid = true;
    return CORB_OK;
}

extern "C" int corb_kf_store_put_from_stereo(CorbKfStore* s, int slot, CorbStereo* sf, int frame, uint64_t id)
{
    int rc = slot_ok(s, slot, "corb_kf_store_put_from_stereo"); if (rc) return rc;
    CorbStereoDeviceFrame f;
    if (corb_stereo_device_frame(sf, frame, &f) != CORB_OK || f.device != s->device) { corb_set_error("corb_kf_store_put_from_stereo: bad front-end / frame / device"); return CORB_ERR_ARG; }
    if (f.cap > s->F) { corb_set_error("corb_kf_store_put_from_stereo: the front-end holds up to %d features per image, the store %d", f.cap, s->F); return CORB_ERR_CAPACITY; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    // on the front-end's stream, behind its run; the store's own stream then waits for the copy
    corb_launch_kf_pack(f.kp, f.desc, f.u_right, f.depth, f.count, -1, id, s->rec(slot), s->F, f.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev, f.stream));
    HIPCHK(hipStreamWaitEvent(s->stream, s->ev, 0));
    s->host[slot].header_valid = false;
    return CORB_OK;
}

static int kf_put_host(CorbKfStore* s, int slot, const CorbKeyPoint* kp, const uint8_t* desc, const float* u_right, const float* depth, int n, uint64_t id, const CorbKeyFrameMeta* meta,
                       const char* who)
{
    int rc = slot_ok(s, slot, who); if (rc) return rc;
    if (n < 0 || n > s->F || (n > 0 && (!kp || !desc)) || (meta && (meta->nlevels < 0 || meta->nlevels > CORB_MAX_LEVELS))) { corb_set_error("%s: bad argument (n = %d, capacity %d)", who, n, s->F); return CORB_ERR_ARG; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    CorbScratch pool(0);
    CorbKeyPoint* dkp; uint8_t* ddesc; float *dur, *ddp; int* dcnt; CorbKeyFrameMeta* dmeta = nullptr;
    std::vector<float> neg((size_t)(n ? n : 1), -1.0f);
    const int cnt = n;
    HIPCHK(pool.upload_block({{(void**)&dkp, kp, (size_t)n * 28}, {(void**)&ddesc, desc, (size_t)n * 32}, {(void**)&dur, u_right ? u_right : neg.data(), (size_t)n * 4},
                              {(void**)&ddp, depth ? depth : neg.data(), (size_t)n * 4}, {(void**)&dcnt, &cnt, 4}, {(void**)&dmeta, meta, meta ? sizeof(CorbKeyFrameMeta) : 0}}));
    corb_launch_kf_pack(dkp, ddesc, dur, ddp, dcnt, n, meta ? meta->id : id, s->rec(slot), s->F, pool.stream);
    if (meta) HIPCHK(hipMemcpyAsync(s->rec(slot) + offsetof(KfHeader, m), dmeta, sizeof(CorbKeyFrameMeta), hipMemcpyDeviceToDevice, pool.stream));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(pool.stream));
    CorbKfStore::Host& h = s->host[slot]; h.n = n; h.n_nodes = 0; h.id = meta ? meta->id : id; h.node_id.clear(); h.header_valid = true;
    return CORB_OK;
}
extern "C" int corb_kf_store_put_host(CorbKfStore* s, int slot, const CorbKeyPoint* kp, const uint8_t* desc, const float* u_right, const float* depth, int n, uint64_t id)
{
    return kf_put_host(s, slot, kp, desc, u_right, depth, n, id, nullptr, "corb_kf_store_put_host");
}
extern "C" int corb_kf_store_put_frame(CorbKfStore* s, int slot, const CorbKeyPoint* kp, const uint8_t* desc, const float* u_right, const float* depth, int n, const CorbKeyFrameMeta* meta)
{
    if (!meta) { corb_set_error("corb_kf_store_put_frame: bad argument"); return CORB_ERR_ARG; }
    return kf_put_host(s, slot, kp, desc, u_right, depth, n, meta->id, meta, "corb_kf_store_put_frame");
}

extern "C" int corb_kf_store_set_bow(CorbKfStore* s, int slot, const CorbFeatVec* fv)
{
    int rc = slot_ok(s, slot, "corb_kf_store_set_bow"); if (rc) return rc;
    if (!fv || fv->n_nodes < 0 || fv->n_nodes > s->F || (fv->n_nodes > 0 && (!fv->node_id || !fv->offset))) { corb_set_error("corb_kf_store_set_bow: bad FeatureVector"); return CORB_ERR_ARG; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    rc = refresh_host(s, slot); if (rc) return rc;
    const int total = fv->n_nodes ? fv->offset[fv->n_nodes] : 0;
    // offsets: start at 0, ascending (the matchers index the descriptor array with them)
    if (fv->n_nodes && fv->offset[0] != 0) { corb_set_error("corb_kf_store_set_bow: offset[0] != 0"); return CORB_ERR_ARG; }
    for (int k = 0; k < fv->n_nodes; k++) if (fv->offset[k] < 0 || fv->offset[k] > fv->offset[k + 1]) { corb_set_error("corb_kf_store_set_bow: offsets not ascending at node %d", k); return CORB_ERR_ARG; }
    if (total < 0 || total > s->F || (total > 0 && !fv->idx)) { corb_set_error("corb_kf_store_set_bow: %d feature indices for a store of %d features per keyframe", total, s->F); return CORB_ERR_ARG; }
    for (int i = 0; i < total; i++) if ((int)fv->idx[i] >= s->host[slot].n) { corb_set_error("corb_kf_store_set_bow: feature index out of range"); return CORB_ERR_ARG; }
    char* r = s->rec(slot);
    if (fv->n_nodes) HIPCHK(hipMemcpyAsync(r + s->L.fv_node, fv->node_id, (size_t)fv->n_nodes * 4, hipMemcpyHostToDevice, s->stream));
    const int zero = 0;
    HIPCHK(hipMemcpyAsync(r + s->L.fv_off, fv->n_nodes ? (const void*)fv->offset : (const void*)&zero, ((size_t)fv->n_nodes + 1) * 4, hipMemcpyHostToDevice, s->stream));
    if (total) HIPCHK(hipMemcpyAsync(r + s->L.fv_idx, fv->idx, (size_t)total * 4, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(r + 4, &fv->n_nodes, 4, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->host[slot].n_nodes = fv->n_nodes; s->host[slot].node_id.assign(fv->node_id, fv->node_id + fv->n_nodes);
    return CORB_OK;
}

extern "C" int corb_kf_store_set_flags(CorbKfStore* s, int slot, const uint8_t* flags)
{
    int rc = slot_ok(s, slot, "corb_kf_store_set_flags"); if (rc) return rc;
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    rc = refresh_host(s, slot); if (rc) return rc;
    const int n = s->host[slot].n;
    if (flags && n) HIPCHK(hipMemcpyAsync(s->rec(slot) + s->L.flags, flags, (size_t)n, hipMemcpyHostToDevice, s->stream));
    else HIPCHK(hipMemsetAsync(s->rec(slot) + s->L.flags, 0, (size_t)s->F, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CORB_OK;
}

extern "C" int corb_kf_store_get(CorbKfStore* s, int slot, CorbKeyPoint* kp, uint8_t* desc, float* u_right, float* depth, uint8_t* flags, int cap, int* n, uint64_t* id,
                                 uint32_t* fv_node_id, int32_t* fv_offset, uint32_t* fv_idx, int32_t* fv_n_nodes)
{
    int rc = slot_ok(s, slot, "corb_kf_store_get"); if (rc) return rc;
    if (!n) return CORB_ERR_ARG;
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    rc = refresh_host(s, slot); if (rc) return rc;
    const CorbKfStore::Host& h = s->host[slot];
    *n = h.n; if (id) *id = h.id; if (fv_n_nodes) *fv_n_nodes = h.n_nodes;
    if (h.n > cap && (kp || desc || u_right || depth || flags)) return CORB_ERR_CAPACITY;
    const char* r = s->rec(slot); const size_t m = (size_t)h.n;
    if (kp && m) HIPCHK(hipMemcpyAsync(kp, r + s->L.kp, m * 28, hipMemcpyDeviceToHost, s->stream));
    if (desc && m) HIPCHK(hipMemcpyAsync(desc, r + s->L.desc, m * 32, hipMemcpyDeviceToHost, s->stream));
    if (u_right && m) HIPCHK(hipMemcpyAsync(u_right, r + s->L.ur, m * 4, hipMemcpyDeviceToHost, s->stream));
    if (depth && m) HIPCHK(hipMemcpyAsync(depth, r + s->L.depth, m * 4, hipMemcpyDeviceToHost, s->stream));
    if (flags && m) HIPCHK(hipMemcpyAsync(flags, r + s->L.flags, m, hipMemcpyDeviceToHost, s->stream));
    if (fv_node_id && h.n_nodes) memcpy(fv_node_id, h.node_id.data(), (size_t)h.n_nodes * 4);
    if (fv_offset) HIPCHK(hipMemcpyAsync(fv_offset, r + s->L.fv_off, ((size_t)h.n_nodes + 1) * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (fv_idx && fv_offset && h.n_nodes && fv_offset[h.n_nodes] > 0 && fv_offset[h.n_nodes] <= s->F) { HIPCHK(hipMemcpyAsync(fv_idx, r + s->L.fv_idx, (size_t)fv_offset[h.n_nodes] * 4, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
    return CORB_OK;
}

// ---- matchers on slots: the kernels of corb_match.cpp on the records' device arrays ----
static void common_nodes(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b, std::vector<int>& pa, std::vector<int>& pb)
{
    size_t i = 0, j = 0;
    while (i < a.size() && j < b.size()) { if (a[i] == b[j]) { pa.push_back((int)i++); pb.push_back((int)j++); } else if (a[i] < b[j]) i++; else j++; }
}
static int two_slots(CorbKfStore* a, int sa, CorbKfStore* b, int sb, const char* who)
{
    int rc = slot_ok(a, sa, who); if (rc) return rc;
    rc = slot_ok(b, sb, who); if (rc) return rc;
    if (a->device != b->device) { corb_set_error("%s: the two stores live on different devices", who); return CORB_ERR_ARG; }
    rc = corb_select_device(a->device); if (rc) return rc;
    { std::lock_guard<std::mutex> lk(a->mu); rc = refresh_host(a, sa); if (rc) return rc; }
    { std::lock_guard<std::mutex> lk(b->mu); rc = refresh_host(b, sb); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(a->stream)); HIPCHK(hipStreamSynchronize(b->stream));      // pending fills of the two records
    return CORB_OK;
}

extern "C" int corb_search_by_bow_slots(int variant, CorbKfStore* A, int sa, CorbKfStore* B, int sb, float nnratio, int check_orientation, int32_t* match, int* n_matches)
{
    if ((variant != 0 && variant != 1) || !n_matches) { corb_set_error("corb_search_by_bow_slots: bad argument"); return CORB_ERR_ARG; }
    int rc = two_slots(A, sa, B, sb, "corb_search_by_bow_slots"); if (rc) return rc;
    const CorbKfStore::Host& ha = A->host[sa]; const CorbKfStore::Host& hb = B->host[sb];
    const int n1 = ha.n, n2 = hb.n, n_out = variant == 0 ? n2 : n1;
    *n_matches = 0;
    if (n_out > 0 && !match) return CORB_ERR_ARG;
    for (int i = 0; i < n_out; i++) match[i] = -1;
    std::vector<int> pa, pb; common_nodes(ha.node_id, hb.node_id, pa, pb);
    if (pa.empty() || n1 == 0 || n2 == 0) return CORB_OK;
    CorbScratch pool(0);
    int *dpa, *dpb, *dmatch, *dbin, *dhist; uint8_t* ones = nullptr;
    HIPCHK(pool.upload_block({{(void**)&dpa, pa.data(), pa.size() * 4}, {(void**)&dpb, pb.data(), pb.size() * 4}}));
    HIPCHK(pool.alloc(&dmatch, (size_t)n_out)); HIPCHK(pool.alloc(&dbin, (size_t)n_out)); HIPCHK(pool.alloc(&dhist, (size_t)CORB_HISTO_LENGTH + 1));
    HIPCHK(hipMemsetAsync(dmatch, 0xFF, (size_t)n_out * 4, pool.stream)); HIPCHK(hipMemsetAsync(dbin, 0xFF, (size_t)n_out * 4, pool.stream));
    HIPCHK(hipMemsetAsync(dhist, 0, (CORB_HISTO_LENGTH + 1) * 4, pool.stream));
    if (variant == 0) { HIPCHK(pool.alloc(&ones, (size_t)n2)); HIPCHK(hipMemsetAsync(ones, 1, (size_t)n2, pool.stream)); }     // the Frame side has no validity test (:212)
    const char* ra = A->rec(sa); const char* rb = B->rec(sb);
    CorbBowDev d;
    d.variant = variant; d.check_ori = check_orientation ? 1 : 0; d.n_pairs = (int)pa.size(); d.nnratio = nnratio;
    d.pair_a = dpa; d.pair_b = dpb;
    d.off1 = (const int*)(ra + A->L.fv_off); d.idx1 = (const int*)(ra + A->L.fv_idx); d.off2 = (const int*)(rb + B->L.fv_off); d.idx2 = (const int*)(rb + B->L.fv_idx);
    d.desc1 = (const unsigned long long*)(ra + A->L.desc); d.desc2 = (const unsigned long long*)(rb + B->L.desc);
    d.angle1 = (const float*)(ra + A->L.angle); d.angle2 = (const float*)(rb + B->L.angle);
    d.valid1 = (const uint8_t*)(ra + A->L.flags); d.valid2 = variant == 1 ? (const uint8_t*)(rb + B->L.flags) : ones;
    d.match = dmatch; d.bin = dbin; d.hist = dhist; d.n_matches = dhist + CORB_HISTO_LENGTH;
    corb_launch_bow(d, n_out, pool.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(match, dmatch, (size_t)n_out * 4, hipMemcpyDeviceToHost, pool.stream));
    HIPCHK(hipMemcpyAsync(n_matches, d.n_matches, 4, hipMemcpyDeviceToHost, pool.stream));
    HIPCHK(hipStreamSynchronize(pool.stream));
    return CORB_OK;
}

extern "C" int corb_search_for_triangulation_slots(CorbKfStore* A, int sa, CorbKfStore* B, int sb, const float* F12, float ex, float ey, const float* scale2,
                                                   const float* sigma2_2, int nlevels, int only_stereo, int check_orientation, int32_t* pairs, int* n_matches)
{
    if (!F12 || !scale2 || !sigma2_2 || nlevels < 1 || nlevels > CORB_MAX_LEVELS || !n_matches) { corb_set_error("corb_search_for_triangulation_slots: bad argument"); return CORB_ERR_ARG; }
    int rc = two_slots(A, sa, B, sb, "corb_search_for_triangulation_slots"); if (rc) return rc;
    const CorbKfStore::Host& ha = A->host[sa]; const CorbKfStore::Host& hb = B->host[sb];
    const int n1 = ha.n, n2 = hb.n;
    *n_matches = 0;
    std::vector<int> pa, pb; common_nodes(ha.node_id, hb.node_id, pa, pb);
    if (pa.empty() || n1 == 0 || n2 == 0) return CORB_OK;
    if (!pairs) return CORB_ERR_ARG;
    // the queries (KF1 features without a MapPoint, stereo if required, :836-847) are built on the device from the record's groups, flags and mvuRight;
    // the host supplies the common vocabulary nodes from its mirror of the node ids.  No read-back before the match kernel.
    const char* ra = A->rec(sa); const char* rb = B->rec(sb);
    CorbScratch pool(0);
    int *dpa, *dpb, *dq1, *dq2, *dmatch, *dbin, *dhist, *dnq; float *dsc, *dsg;
    // the initial state of the outputs (-1 matches and bins, zero histogram and counters) travels in the same host-to-device block as the inputs:
    // one copy instead of four memset launches
    static thread_local std::vector<int> init;
    const size_t n_init = 2 * (size_t)n1 + CORB_HISTO_LENGTH + 2;
    if (init.size() < n_init) init.resize(n_init);
    std::fill(init.begin(), init.begin() + 2 * (size_t)n1, -1); std::fill(init.begin() + 2 * (size_t)n1, init.begin() + n_init, 0);
    int* dinit;
    HIPCHK(pool.upload_block({{(void**)&dpa, pa.data(), pa.size() * 4}, {(void**)&dpb, pb.data(), pb.size() * 4},
                              {(void**)&dsc, scale2, (size_t)nlevels * 4}, {(void**)&dsg, sigma2_2, (size_t)nlevels * 4}, {(void**)&dinit, init.data(), n_init * 4}}));
    dmatch = dinit; dbin = dinit + n1; dhist = dinit + 2 * (size_t)n1; dnq = dhist + CORB_HISTO_LENGTH + 1;      // hist[HISTO_LENGTH] = n_matches
    HIPCHK(pool.alloc(&dq1, (size_t)n1)); HIPCHK(pool.alloc(&dq2, (size_t)n1));
    HIPCHK(hipStreamSynchronize(A->stream)); if (B != A) HIPCHK(hipStreamSynchronize(B->stream));      // records are written on the stores' streams
    corb_launch_tri_queries((const int*)(ra + A->L.fv_off), (const int*)(ra + A->L.fv_idx), (const uint8_t*)(ra + A->L.flags), (const float*)(ra + A->L.ur),
                            dpa, dpb, (int)pa.size(), only_stereo ? 1 : 0, dq1, dq2, dnq, pool.stream);
    CorbTriDev d;
    d.n_queries = n1; d.n_queries_dev = dnq; d.only_stereo = only_stereo ? 1 : 0; d.check_ori = check_orientation ? 1 : 0;
    d.q_idx1 = dq1; d.q_node2 = dq2; d.off2 = (const int*)(rb + B->L.fv_off); d.idx2 = (const int*)(rb + B->L.fv_idx);
    d.desc1 = (const unsigned long long*)(ra + A->L.desc); d.desc2 = (const unsigned long long*)(rb + B->L.desc);
    d.kp1 = (const CorbKeyPoint*)(ra + A->L.kp); d.kp2 = (const CorbKeyPoint*)(rb + B->L.kp);
    d.uright1 = (const float*)(ra + A->L.ur); d.uright2 = (const float*)(rb + B->L.ur);
    d.has_mp2 = (const uint8_t*)(rb + B->L.flags);
    for (int i = 0; i < 9; i++) d.F12[i] = F12[i];
    d.ex = ex; d.ey = ey; d.scale2 = dsc; d.sigma2_2 = dsg;
    d.match = dmatch; d.bin = dbin; d.hist = dhist; d.n_matches = dhist + CORB_HISTO_LENGTH;
    corb_launch_tri(d, n1, pool.stream);
    HIPCHK(hipGetLastError());
    static thread_local std::vector<int> m12; m12.resize((size_t)n1 + 1);
    HIPCHK(pool.d2h(m12.data(), dmatch, (size_t)n1 * 4));      // (through the lane's page-locked staging: enqueued, not a blocking pageable copy)
    HIPCHK(pool.d2h(n_matches, d.n_matches, 4));
    HIPCHK(pool.fetch_finish());
    int k = 0;
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) { pairs[2 * k] = i; pairs[2 * k + 1] = m12[i]; k++; }
    return CORB_OK;
}


// ---- the rest of the keyframe payload: pose, intrinsics, GBA fields, per-feature map-point ids (KeyFrame.h:65-79) ----
extern "C" int corb_kf_store_set_meta(CorbKfStore* s, int slot, const CorbKeyFrameMeta* meta)
{
    int rc = slot_ok(s, slot, "corb_kf_store_set_meta"); if (rc) return rc;
    if (!meta || meta->nlevels < 0 || meta->nlevels > CORB_MAX_LEVELS) { corb_set_error("corb_kf_store_set_meta: bad argument"); return CORB_ERR_ARG; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipMemcpyAsync(s->rec(slot) + offsetof(KfHeader, m), meta, sizeof(CorbKeyFrameMeta), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->host[slot].header_valid) s->host[slot].id = meta->id;
    return CORB_OK;
}
extern "C" int corb_kf_store_get_meta(CorbKfStore* s, int slot, CorbKeyFrameMeta* meta)
{
    int rc = slot_ok(s, slot, "corb_kf_store_get_meta"); if (rc) return rc;
    if (!meta) return CORB_ERR_ARG;
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipMemcpyAsync(meta, s->rec(slot) + offsetof(KfHeader, m), sizeof(CorbKeyFrameMeta), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CORB_OK;
}
extern "C" int corb_kf_store_set_map_points(CorbKfStore* s, int slot, const uint64_t* mp_id)
{
    int rc = slot_ok(s, slot, "corb_kf_store_set_map_points"); if (rc) return rc;
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    rc = refresh_host(s, slot); if (rc) return rc;
    const int n = s->host[slot].n;
    if (n > 0 && !mp_id) { corb_set_error("corb_kf_store_set_map_points: NULL ids"); return CORB_ERR_ARG; }
    if (n > 0) HIPCHK(hipMemcpyAsync(s->rec(slot) + s->L.mp_id, mp_id, (size_t)n * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CORB_OK;
}
extern "C" int corb_kf_store_get_map_points(CorbKfStore* s, int slot, uint64_t* mp_id, int cap)
{
    int rc = slot_ok(s, slot, "corb_kf_store_get_map_points"); if (rc) return rc;
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    rc = refresh_host(s, slot); if (rc) return rc;
    const int n = s->host[slot].n;
    if (n > cap || (n > 0 && !mp_id)) return CORB_ERR_CAPACITY;
    if (n > 0) HIPCHK(hipMemcpyAsync(mp_id, s->rec(slot) + s->L.mp_id, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CORB_OK;
}

extern "C" int corb_kf_store_put_batch(CorbKfStore* s, int first, int n, const CorbKeyFrameMeta* meta, const int32_t* feat_offset, const CorbKeyPoint* kp, const uint8_t* desc,
                                       const float* u_right, const float* depth, const uint64_t* mp_id)
{
    if (!s || first < 0 || n < 0 || (long long)first + n > s->capacity) { corb_set_error("corb_kf_store_put_batch: bad store / slot range"); return CORB_ERR_ARG; }
    if (n == 0) return CORB_OK;
    if (!meta || !feat_offset || feat_offset[0] != 0) { corb_set_error("corb_kf_store_put_batch: bad argument"); return CORB_ERR_ARG; }
    for (int i = 0; i < n; i++) {
        const int c = feat_offset[i + 1] - feat_offset[i];
        if (c < 0) { corb_set_error("corb_kf_store_put_batch: offsets not ascending"); return CORB_ERR_ARG; }
        if (c > s->F) { corb_set_error("corb_kf_store_put_batch: keyframe %d has %d features, the store holds %d per keyframe", i, c, s->F); return CORB_ERR_CAPACITY; }
        if (meta[i].nlevels < 0 || meta[i].nlevels > CORB_MAX_LEVELS) { corb_set_error("corb_kf_store_put_batch: keyframe %d: bad nlevels", i); return CORB_ERR_ARG; }
    }
    const size_t total = (size_t)feat_offset[n];
    if (total > 0 && !kp) { corb_set_error("corb_kf_store_put_batch: NULL keypoints"); return CORB_ERR_ARG; }
    int rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_meta = (size_t)n * sizeof(CorbKeyFrameMeta), b_off = ((size_t)n + 1) * 4, b_kp = total * 28, b_desc = desc ? total * 32 : 0, b_ur = u_right ? total * 4 : 0,
                 b_dp = depth ? total * 4 : 0, b_id = mp_id ? total * 8 : 0;
    char* stage = nullptr;
    HIPCHK(hipMalloc((void**)&stage, al(b_meta) + al(b_off) + al(b_kp) + al(b_desc) + al(b_ur) + al(b_dp) + al(b_id) + 256));
    struct Guard { char* p; ~Guard() { (void)hipFree(p); } } guard{stage};
    size_t o = 0;
    auto put = [&](const void* src, size_t bytes) -> char* { char* d = stage + o; o += al(bytes); if (bytes && hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, s->stream) != hipSuccess) return nullptr; return bytes ? d : nullptr; };
    char* dmeta = put(meta, b_meta); char* doff = put(feat_offset, b_off); char* dkp = put(kp, b_kp); char* ddesc = put(desc, b_desc);
    char* dur = put(u_right, b_ur); char* ddp = put(depth, b_dp); char* did = put(mp_id, b_id);
    if (!dmeta || !doff || (total && !dkp)) { corb_set_error("corb_kf_store_put_batch: upload failed"); return CORB_ERR_HIP; }
    corb_launch_kf_pack_batch((const CorbKeyFrameMeta*)dmeta, (const int*)doff, (const CorbKeyPoint*)dkp, (const uint8_t*)ddesc, (const float*)dur, (const float*)ddp,
                              (const unsigned long long*)did, n, s->base, first, s->F, s->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int i = 0; i < n; i++) { CorbKfStore::Host& h = s->host[first + i]; h.n = feat_offset[i + 1] - feat_offset[i]; h.n_nodes = 0; h.id = meta[i].id; h.node_id.clear(); h.header_valid = true; }
    return CORB_OK;
}

// ---- map-point store (MapPoint.h:52-72) ----
extern "C" int corb_mp_store_create(int device, int capacity, int max_obs, CorbMpStore** out)
{
    if (!out || capacity < 1 || max_obs < 1 || max_obs > 4096) { corb_set_error("corb_mp_store_create: bad argument"); return CORB_ERR_ARG; }
    *out = nullptr;
    int rc = corb_select_device(device); if (rc) return rc;
    CorbMpStore* s = new CorbMpStore();
    s->device = device; s->capacity = capacity; s->O = max_obs; s->L = MpLayout(max_obs);
    if (hipMalloc((void**)&s->base, (size_t)capacity * s->L.bytes) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        corb_set_error("corb_mp_store_create: %d map points x %zu bytes: allocation failed", capacity, s->L.bytes);
        if (s->base) (void)hipFree(s->base);
        delete s; return CORB_ERR_HIP;
    }
    // an empty slot reads as a bad map point without observations
    (void)hipMemsetAsync(s->base, 0, (size_t)capacity * s->L.bytes, s->stream);
    (void)hipStreamSynchronize(s->stream);
    *out = s;
    return CORB_OK;
}
extern "C" void corb_mp_store_destroy(CorbMpStore* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->base) (void)hipFree(s->base);
    if (s->idt.keys) (void)hipFree(s->idt.keys);
    if (s->lba_dev) (void)hipFree(s->lba_dev);
    if (s->lba_host) (void)hipHostFree(s->lba_host);
    if (s->lba_event) (void)hipEventDestroy(s->lba_event);
    delete s;
}
extern "C" int corb_mp_store_record_bytes(const CorbMpStore* s) { return s ? (int)s->L.bytes : 0; }

static int mp_range_ok(CorbMpStore* s, int first, int n, const char* who)
{
    if (!s || first < 0 || n < 0 || (long long)first + n > s->capacity) { corb_set_error("%s: bad store / slot range", who); return CORB_ERR_ARG; }
    return CORB_OK;
}
extern "C" int corb_mp_store_put_host(CorbMpStore* s, int first, int n, const CorbMapPointRecord* records, const int32_t* obs_offset, const uint64_t* obs_kf_id, const uint32_t* obs_feature_idx)
{
    int rc = mp_range_ok(s, first, n, "corb_mp_store_put_host"); if (rc) return rc;
    if (n == 0) return CORB_OK;
    if (!records || !obs_offset || obs_offset[0] != 0) { corb_set_error("corb_mp_store_put_host: bad argument"); return CORB_ERR_ARG; }
    for (int i = 0; i < n; i++) {
        const int c = obs_offset[i + 1] - obs_offset[i];
        if (c < 0) { corb_set_error("corb_mp_store_put_host: offsets not ascending"); return CORB_ERR_ARG; }
        if (c > s->O) { corb_set_error("corb_mp_store_put_host: map point %d has %d observations, the store holds %d per point", i, c, s->O); return CORB_ERR_CAPACITY; }
    }
    const size_t total = (size_t)obs_offset[n];
    if (total > 0 && (!obs_kf_id || !obs_feature_idx)) { corb_set_error("corb_mp_store_put_host: NULL observations"); return CORB_ERR_ARG; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    // staging memory of this call (a batch can be gigabytes: not taken from the per-device arena, which never shrinks)
    const size_t b_hdr = (size_t)n * sizeof(CorbMapPointRecord), b_off = ((size_t)n + 1) * 4, b_kf = total * 8, b_idx = total * 4;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char* stage = nullptr;
    HIPCHK(hipMalloc((void**)&stage, al(b_hdr) + al(b_off) + al(b_kf) + al(b_idx) + 256));
    struct Guard { char* p; ~Guard() { (void)hipFree(p); } } guard{stage};
    CorbMapPointRecord* dh = (CorbMapPointRecord*)stage; int* doff = (int*)(stage + al(b_hdr));
    unsigned long long* dkf = (unsigned long long*)(stage + al(b_hdr) + al(b_off)); uint32_t* didx = (uint32_t*)(stage + al(b_hdr) + al(b_off) + al(b_kf));
    int* dstat = (int*)(stage + al(b_hdr) + al(b_off) + al(b_kf) + al(b_idx));
    HIPCHK(hipMemcpyAsync(dh, records, b_hdr, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(doff, obs_offset, b_off, hipMemcpyHostToDevice, s->stream));
    if (total) { HIPCHK(hipMemcpyAsync(dkf, obs_kf_id, b_kf, hipMemcpyHostToDevice, s->stream)); HIPCHK(hipMemcpyAsync(didx, obs_feature_idx, b_idx, hipMemcpyHostToDevice, s->stream)); }
    HIPCHK(hipMemsetAsync(dstat, 0, 4, s->stream));
    s->idt_valid = false;                                   // the slots may hold other ids now: corb_mp_store_build_index again before the tracking calls
    corb_launch_mp_pack(dh, doff, dkf, didx, n, s->base, first, s->O, dstat, s->stream);
    HIPCHK(hipGetLastError());
    int hstat = 0;                                          // the kernel's own range check (the offsets were validated above: it fires only if the caller's arrays changed under the call)
    HIPCHK(hipMemcpyAsync(&hstat, dstat, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (hstat) { corb_set_error("corb_mp_store_put_host: an observation count left [0, %d] while the records were packed; the skipped records keep their old contents", s->O); return CORB_ERR_CAPACITY; }
    return CORB_OK;
}
extern "C" int corb_mp_store_get(CorbMpStore* s, int first, int n, CorbMapPointRecord* records, uint64_t* obs_kf_id, uint32_t* obs_feature_idx)
{
    int rc = mp_range_ok(s, first, n, "corb_mp_store_get"); if (rc) return rc;
    if (n == 0) return CORB_OK;
    if (!records) return CORB_ERR_ARG;
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    const size_t b_hdr = (size_t)n * sizeof(CorbMapPointRecord), b_kf = obs_kf_id ? (size_t)n * s->O * 8 : 0, b_idx = obs_feature_idx ? (size_t)n * s->O * 4 : 0;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    char* stage = nullptr;
    HIPCHK(hipMalloc((void**)&stage, al(b_hdr) + al(b_kf) + al(b_idx) + 256));
    struct Guard { char* p; ~Guard() { (void)hipFree(p); } } guard{stage};
    CorbMapPointRecord* dh = (CorbMapPointRecord*)stage;
    unsigned long long* dkf = obs_kf_id ? (unsigned long long*)(stage + al(b_hdr)) : nullptr; uint32_t* didx = obs_feature_idx ? (uint32_t*)(stage + al(b_hdr) + al(b_kf)) : nullptr;
    corb_launch_mp_unpack(s->base, first, n, s->O, dh, dkf, didx, s->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(records, dh, b_hdr, hipMemcpyDeviceToHost, s->stream));
    if (dkf) HIPCHK(hipMemcpyAsync(obs_kf_id, dkf, b_kf, hipMemcpyDeviceToHost, s->stream));
    if (didx) HIPCHK(hipMemcpyAsync(obs_feature_idx, didx, b_idx, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CORB_OK;
}

// ---- MapFusion::insertServerMapToGlobleMap on records (S/src/MapFusion.cpp:622-658) ----
extern "C" int corb_rebase_map_store(const float* To2n, CorbKfStore* kf, const int32_t* kf_slots, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp)
{
    if (!To2n || n_kf < 0 || n_mp < 0 || (n_kf > 0 && (!kf || !kf_slots)) || (n_mp > 0 && (!mp || !mp_slots))) { corb_set_error("corb_rebase_map_store: bad argument"); return CORB_ERR_ARG; }
    if (kf && mp && kf->device != mp->device) { corb_set_error("corb_rebase_map_store: the stores live on different devices"); return CORB_ERR_ARG; }
    for (int i = 0; i < n_kf; i++) if (kf_slots[i] < 0 || kf_slots[i] >= kf->capacity) { corb_set_error("corb_rebase_map_store: keyframe slot out of range"); return CORB_ERR_ARG; }
    for (int i = 0; i < n_mp; i++) if (mp_slots[i] < 0 || mp_slots[i] >= mp->capacity) { corb_set_error("corb_rebase_map_store: map point slot out of range"); return CORB_ERR_ARG; }
    if (n_kf == 0 && n_mp == 0) return CORB_OK;
    int rc = corb_select_device(kf ? kf->device : mp->device); if (rc) return rc;
    // the stores' documented lock order (keyframes, then map points), held until the re-based records are complete: a solve, a tracking call or the packing /
    // receiving phase of a push from another host thread sees the map either before or after the re-basing.  NOT covered: the window of an asynchronous push
    // between corb_map_push_begin and _wait, during which the records in flight belong to the push (include/corb_accel.h) -- the caller keeps re-basing out of it.
    std::unique_lock<std::mutex> lk_kf, lk_mp;
    if (kf) lk_kf = std::unique_lock<std::mutex>(kf->mu);
    if (mp) lk_mp = std::unique_lock<std::mutex>(mp->mu);
    if (kf) HIPCHK(hipStreamSynchronize(kf->stream));
    if (mp) HIPCHK(hipStreamSynchronize(mp->stream));
    CorbScratch pool(0);
    float* dT; int *dks, *dms;
    HIPCHK(pool.upload_block({{(void**)&dT, To2n, 64}, {(void**)&dks, kf_slots, (size_t)n_kf * 4}, {(void**)&dms, mp_slots, (size_t)n_mp * 4}}));
    corb_launch_rebase_records(dT, kf ? kf->base : nullptr, kf ? kf->L.bytes : 0, dks, n_kf, mp ? mp->base : nullptr, mp ? mp->L.bytes : 0, dms, n_mp, pool.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(pool.stream));
    return CORB_OK;
}

// ---- MapPoint counters / MapPoint::Replace on records (include/corb_accel.h) ----
void corb_launch_mp_replace(char* mp_base, size_t mp_bytes, int max_obs, int slot_this, int slot_into, char* kf_base, size_t kf_bytes, int F, int kf_first, int kf_n,
                            CorbIdTable kfid, unsigned long long* desc, int* status, hipStream_t s);
void corb_launch_mp_counters(char* base, size_t bytes, int first, int n, CorbMapPointCounters* io, int set, hipStream_t s);
static_assert(sizeof(CorbMapPointRecord) + sizeof(CorbMapPointCounters) == CORB_MP_HEADER_BYTES, "the counters fill the header's spare bytes");

static int mp_counters(CorbMpStore* s, int first, int n, CorbMapPointCounters* host, int set, const char* who)
{
    int rc = mp_range_ok(s, first, n, who); if (rc) return rc;
    if (n == 0) return CORB_OK;
    if (!host) { corb_set_error("%s: NULL array", who); return CORB_ERR_ARG; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    CorbMapPointCounters* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)n * sizeof(CorbMapPointCounters) + 256));
    struct Guard { void* p; ~Guard() { (void)hipFree(p); } } guard{d};
    if (set) HIPCHK(hipMemcpyAsync(d, host, (size_t)n * sizeof(CorbMapPointCounters), hipMemcpyHostToDevice, s->stream));
    corb_launch_mp_counters(s->base, s->L.bytes, first, n, d, set, s->stream);
    HIPCHK(hipGetLastError());
    if (!set) HIPCHK(hipMemcpyAsync(host, d, (size_t)n * sizeof(CorbMapPointCounters), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CORB_OK;
}
void corb_launch_mp_scratch(char* base, size_t bytes, size_t off, int first, int n, void* io, int set, hipStream_t s);
static int mp_scratch(CorbMpStore* s, int first, int n, CorbMapPointScratch* host, int set, const char* who)
{
    int rc = mp_range_ok(s, first, n, who); if (rc) return rc;
    if (n == 0) return CORB_OK;
    if (!host) { corb_set_error("%s: NULL array", who); return CORB_ERR_ARG; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    char* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, (size_t)n * sizeof(CorbMapPointScratch) + 256));
    struct Guard { void* p; ~Guard() { (void)hipFree(p); } } guard{d};
    if (set) HIPCHK(hipMemcpyAsync(d, host, (size_t)n * sizeof(CorbMapPointScratch), hipMemcpyHostToDevice, s->stream));
    corb_launch_mp_scratch(s->base, s->L.bytes, s->L.scratch, first, n, d, set, s->stream);
    HIPCHK(hipGetLastError());
    if (!set) HIPCHK(hipMemcpyAsync(host, d, (size_t)n * sizeof(CorbMapPointScratch), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return CORB_OK;
}
extern "C" int corb_mp_store_set_scratch(CorbMpStore* s, int first, int n, const CorbMapPointScratch* c) { return mp_scratch(s, first, n, const_cast<CorbMapPointScratch*>(c), 1, "corb_mp_store_set_scratch"); }
extern "C" int corb_mp_store_get_scratch(CorbMpStore* s, int first, int n, CorbMapPointScratch* c) { return mp_scratch(s, first, n, c, 0, "corb_mp_store_get_scratch"); }
extern "C" int corb_mp_store_set_counters(CorbMpStore* s, int first, int n, const CorbMapPointCounters* c) { return mp_counters(s, first, n, const_cast<CorbMapPointCounters*>(c), 1, "corb_mp_store_set_counters"); }
extern "C" int corb_mp_store_get_counters(CorbMpStore* s, int first, int n, CorbMapPointCounters* c) { return mp_counters(s, first, n, c, 0, "corb_mp_store_get_counters"); }

extern "C" int corb_mp_store_replace(CorbMpStore* map, int slot_this, int slot_into, CorbKfStore* kf, int kf_first, int kf_n, int* status)
{
    if (!map || !kf || slot_this < 0 || slot_this >= map->capacity || slot_into < 0 || slot_into >= map->capacity || kf_first < 0 || kf_n < 0 || (long long)kf_first + kf_n > kf->capacity) {
        corb_set_error("corb_mp_store_replace: bad store / slot"); return CORB_ERR_ARG;
    }
    if (kf->device != map->device) { corb_set_error("corb_mp_store_replace: the stores live on different devices"); return CORB_ERR_ARG; }
    if (map->O > 1024) { corb_set_error("corb_mp_store_replace: more than 1024 observations per map point"); return CORB_ERR_ARG; }
    if (status) *status = 0;
    if (slot_this == slot_into) { if (status) *status = 1; return CORB_OK; }
    int rc = corb_select_device(map->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(kf->mu); std::lock_guard<std::mutex> lk2(map->mu);       // (always in this order)
    HIPCHK(hipStreamSynchronize(kf->stream)); HIPCHK(hipStreamSynchronize(map->stream));
    CorbScratch pool(0);
    CorbIdTable kfid; unsigned int cap = 64; while (cap < 2u * (unsigned int)(kf_n > 0 ? kf_n : 1)) cap <<= 1;
    HIPCHK(pool.alloc(&kfid.keys, (size_t)cap)); HIPCHK(pool.alloc(&kfid.vals, (size_t)cap)); kfid.mask = cap - 1;
    HIPCHK(hipMemsetAsync(kfid.keys, 0xFF, (size_t)cap * 8, pool.stream));
    HIPCHK(hipMemsetAsync(kfid.vals, 0x7F, (size_t)cap * 4, pool.stream));               // (kf_index_kernel: corb_idtab_insert_min)
    unsigned long long* desc = nullptr; int* dst = nullptr;
    HIPCHK(pool.alloc(&desc, (size_t)map->O * 4)); HIPCHK(pool.alloc(&dst, 1));
    HIPCHK(hipMemsetAsync(dst, 0, 4, pool.stream));
    corb_launch_mp_replace(map->base, map->L.bytes, map->O, slot_this, slot_into, kf->base, kf->L.bytes, kf->F, kf_first, kf_n, kfid, desc, dst, pool.stream);
    HIPCHK(hipGetLastError());
    int* res = static_cast<int*>(pool.pinned());
    HIPCHK(hipMemcpyAsync(res, dst, 4, hipMemcpyDeviceToHost, pool.stream));
    HIPCHK(pool.fetch_finish());
    if (res[0] == CORB_ERR_CAPACITY) { corb_set_error("corb_mp_store_replace: the target map point has no room for the observations (max_observations = %d); nothing was written", map->O); return CORB_ERR_CAPACITY; }
    if (status) *status = res[0];
    return CORB_OK;
}
