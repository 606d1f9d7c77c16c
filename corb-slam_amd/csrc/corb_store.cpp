// corb_store.cpp -- device-resident keyframe store, matchers on store slots, and the RCCL map push (see include/corb_accel.h).
// Replaces, for the hot path, what the reference moves as boost text archives through ROS services: the per-keyframe payload of
// corbslam_client/include/KeyFrame.h:59-87 (keypoints, descriptors, mvuRight, mvDepth, FeatureVector) and the client -> server batch of
// corbslam_client/src/Cache.cc:322-375 / DataDriver.cc:135-193.  One fixed-size SoA record per keyframe in HBM; a push is ncclSend / ncclRecv of
// whole records between the ranks' device buffers (xGMI), no serialisation, no host staging.
#include "match_internal.h"
#include "store_internal.h"
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

struct CorbKfStore {
    int device = 0, capacity = 0, F = 0;
    RecLayout L{1};
    char* base = nullptr;                     // [capacity][L.bytes]
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    std::mutex mu;
    struct Host { int n = -1, n_nodes = 0; unsigned long long id = 0; std::vector<uint32_t> node_id; bool header_valid = false; };
    std::vector<Host> host;                   // host mirror of the small parts (counts, vocabulary node ids)
    char* rec(int slot) const { return base + (size_t)slot * L.bytes; }
};

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
    if (h.n_nodes) { HIPCHK(hipMemcpyAsync(h.node_id.data(), s->rec(slot) + s->L.fv_node, (size_t)h.n_nodes * 4, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
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

extern "C" int corb_kf_store_put_host(CorbKfStore* s, int slot, const CorbKeyPoint* kp, const uint8_t* desc, const float* u_right, const float* depth, int n, uint64_t id)
{
    int rc = slot_ok(s, slot, "corb_kf_store_put_host"); if (rc) return rc;
    if (n < 0 || n > s->F || (n > 0 && (!kp || !desc))) { corb_set_error("corb_kf_store_put_host: bad argument (n = %d, capacity %d)", n, s->F); return CORB_ERR_ARG; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    CorbScratch pool(0);
    CorbKeyPoint* dkp; uint8_t* ddesc; float *dur, *ddp; int* dcnt;
    std::vector<float> neg((size_t)(n ? n : 1), -1.0f);
    const int cnt = n;
    HIPCHK(pool.upload_block({{(void**)&dkp, kp, (size_t)n * 28}, {(void**)&ddesc, desc, (size_t)n * 32}, {(void**)&dur, u_right ? u_right : neg.data(), (size_t)n * 4},
                              {(void**)&ddp, depth ? depth : neg.data(), (size_t)n * 4}, {(void**)&dcnt, &cnt, 4}}));
    corb_launch_kf_pack(dkp, ddesc, dur, ddp, dcnt, n, id, s->rec(slot), s->F, pool.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(pool.stream));
    CorbKfStore::Host& h = s->host[slot]; h.n = n; h.n_nodes = 0; h.id = id; h.node_id.clear(); h.header_valid = true;
    return CORB_OK;
}

extern "C" int corb_kf_store_set_bow(CorbKfStore* s, int slot, const CorbFeatVec* fv)
{
    int rc = slot_ok(s, slot, "corb_kf_store_set_bow"); if (rc) return rc;
    if (!fv || fv->n_nodes < 0 || fv->n_nodes > s->F || (fv->n_nodes > 0 && (!fv->node_id || !fv->offset))) { corb_set_error("corb_kf_store_set_bow: bad FeatureVector"); return CORB_ERR_ARG; }
    rc = corb_select_device(s->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    rc = refresh_host(s, slot); if (rc) return rc;
    const int total = fv->n_nodes ? fv->offset[fv->n_nodes] : 0;
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
    if (fv_idx && fv_offset && h.n_nodes && fv_offset[h.n_nodes] > 0) { HIPCHK(hipMemcpyAsync(fv_idx, r + s->L.fv_idx, (size_t)fv_offset[h.n_nodes] * 4, hipMemcpyDeviceToHost, s->stream)); HIPCHK(hipStreamSynchronize(s->stream)); }
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
    std::vector<int> m12(n1);
    HIPCHK(hipMemcpyAsync(m12.data(), dmatch, (size_t)n1 * 4, hipMemcpyDeviceToHost, pool.stream));
    HIPCHK(hipMemcpyAsync(n_matches, d.n_matches, 4, hipMemcpyDeviceToHost, pool.stream));
    HIPCHK(hipStreamSynchronize(pool.stream));
    int k = 0;
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) { pairs[2 * k] = i; pairs[2 * k + 1] = m12[i]; k++; }
    return CORB_OK;
}

// ---- RCCL (librccl.so loaded on first use: a process that never pushes a map does not pay for it) ----
namespace {
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId_ { char internal[128]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId_*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl& rccl()
{
    static Rccl r; static std::once_flag once;
    std::call_once(once, [] {
        // the RCCL that belongs to the HIP runtime THIS library is linked with (its directory): a process may hold a second copy of the ROCm libraries
        // (a Python framework's bundled ones), and streams / events of one runtime mean nothing to the other
        std::string own;
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
            own = info.dli_fname;
            const size_t slash = own.rfind('/');
            own = slash == std::string::npos ? std::string() : own.substr(0, slash + 1) + "librccl.so";
        }
        for (const char* name : {own.c_str(), "/opt/rocm/lib/librccl.so", "librccl.so", "librccl.so.1"}) { if (!*name) continue; r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
        if (!r.lib) return;
        auto sym = [&](const char* n) { return dlsym(r.lib, n); };
        r.GetUniqueId = (int (*)(ncclUniqueId_*))sym("ncclGetUniqueId"); r.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId_, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy"); r.Send = (int (*)(const void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclSend");
        r.Recv = (int (*)(void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclRecv");
        r.AllGather = (int (*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))sym("ncclAllGather");
        r.GroupStart = (int (*)())sym("ncclGroupStart"); r.GroupEnd = (int (*)())sym("ncclGroupEnd"); r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.AllGather && r.GroupStart && r.GroupEnd;
    });
    return r;
}
const int NCCL_INT8 = 0, NCCL_INT32 = 2;      // ncclDataType_t: ncclInt8 = 0 (= ncclChar), ncclInt32 = 2 (rccl.h)
}
#define NCCLCHK(call) do { int e_ = (call); if (e_ != 0) { corb_set_error("%s failed: %s", #call, rccl().GetErrorString ? rccl().GetErrorString(e_) : "rccl error"); return CORB_ERR_HIP; } } while (0)

struct CorbComm { ncclComm_t comm = nullptr; int rank = 0, world = 1, device = 0; hipStream_t stream = nullptr; int* d_counts = nullptr; };

extern "C" int corb_comm_unique_id(void* id128)
{
    if (!id128) return CORB_ERR_ARG;
    if (!rccl().ok) { corb_set_error("librccl.so could not be loaded"); return CORB_ERR_HIP; }
    ncclUniqueId_ id; NCCLCHK(rccl().GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return CORB_OK;
}
extern "C" int corb_comm_create(const void* id128, int rank, int world, int device, CorbComm** out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) { corb_set_error("corb_comm_create: bad argument"); return CORB_ERR_ARG; }
    *out = nullptr;
    if (!rccl().ok) { corb_set_error("librccl.so could not be loaded"); return CORB_ERR_HIP; }
    int rc = corb_select_device(device); if (rc) return rc;
    CorbComm* c = new CorbComm(); c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId_ id; memcpy(id.internal, id128, 128);
    if (rccl().CommInitRank(&c->comm, world, id, rank) != 0) { corb_set_error("ncclCommInitRank failed (rank %d of %d)", rank, world); delete c; return CORB_ERR_HIP; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&c->d_counts, sizeof(int) * ((size_t)world + 1)) != hipSuccess) {
        corb_set_error("corb_comm_create: stream / buffer allocation failed"); (void)rccl().CommDestroy(c->comm); delete c; return CORB_ERR_HIP;
    }
    *out = c;
    return CORB_OK;
}
extern "C" void corb_comm_destroy(CorbComm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->d_counts) (void)hipFree(c->d_counts);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    delete c;
}

extern "C" int corb_map_push(CorbComm* c, CorbKfStore* s, const int* slots, int n_slots, int root, const int* dst_first, int* recv_counts)
{
    if (!c || !s || n_slots < 0 || (n_slots > 0 && !slots) || root < 0 || root >= c->world || c->device != s->device) { corb_set_error("corb_map_push: bad argument"); return CORB_ERR_ARG; }
    for (int i = 0; i < n_slots; i++) if (slots[i] < 0 || slots[i] >= s->capacity) { corb_set_error("corb_map_push: slot %d out of range", slots[i]); return CORB_ERR_ARG; }
    if (c->rank == root && !dst_first) { corb_set_error("corb_map_push: the root needs dst_first[world]"); return CORB_ERR_ARG; }
    int rc = corb_select_device(c->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk(s->mu);
    HIPCHK(hipStreamSynchronize(s->stream));                       // pending fills of the records that are about to travel
    // 1. how many keyframes every rank sends (one int each, all-gather on the device)
    HIPCHK(hipMemcpyAsync(c->d_counts + c->world, &n_slots, sizeof(int), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(rccl().AllGather(c->d_counts + c->world, c->d_counts, 1, NCCL_INT32, c->comm, c->stream));
    std::vector<int> counts(c->world);
    HIPCHK(hipMemcpyAsync(counts.data(), c->d_counts, sizeof(int) * c->world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->rank == root)
        for (int r = 0; r < c->world; r++) if (counts[r] < 0 || dst_first[r] < 0 || dst_first[r] + counts[r] > s->capacity) { corb_set_error("corb_map_push: rank %d sends %d keyframes, no room at slot %d", r, counts[r], dst_first[r]); return CORB_ERR_CAPACITY; }
    // 2. the records: one grouped exchange, device buffer to device buffer
    const size_t B = s->L.bytes;
    NCCLCHK(rccl().GroupStart());
    for (int i = 0; i < n_slots; i++) NCCLCHK(rccl().Send(s->rec(slots[i]), B, NCCL_INT8, root, c->comm, c->stream));
    if (c->rank == root)
        for (int r = 0; r < c->world; r++)
            for (int i = 0; i < counts[r]; i++) NCCLCHK(rccl().Recv(s->rec(dst_first[r] + i), B, NCCL_INT8, r, c->comm, c->stream));
    NCCLCHK(rccl().GroupEnd());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->rank == root) {
        for (int r = 0; r < c->world; r++) for (int i = 0; i < counts[r]; i++) s->host[dst_first[r] + i].header_valid = false;
        if (recv_counts) memcpy(recv_counts, counts.data(), sizeof(int) * c->world);
    }
    return CORB_OK;
}
