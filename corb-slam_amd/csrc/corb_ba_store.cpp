// corb_ba_store.cpp -- Optimizer::GlobalBundleAdjustemnt on store records (see include/corb_accel.h: corb_ba_solve_store).
// What the server rank runs after a map push and the re-basing (corbslam_server/src/GlobalOptimize.cpp:435-547 -> corbslam_client/src/Optimizer.cc:43-270):
// the graph is derived on the device from the keyframe / map-point records, solved, and the estimates are written back into the records.
#include "store_host.h"
#include "ba_store_internal.h"
static_assert(sizeof(CorbBAOptions) == 32, "CorbBAOptions: scale_factor fills what was padding -- the struct's size is part of the C-ABI");
#include "ba_device_problem.h"
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <thread>
#include <atomic>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);
bool corb_ba_staged_device_wanted(const CorbBAStage* stages, int n_stages);
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

namespace {
struct Lap {                          // CORB_BA_TIMING=1: host-side phase times of a call on stderr (development aid, as in corb_ba.cpp)
    bool on; std::chrono::steady_clock::time_point t;
    Lap() : on(getenv("CORB_BA_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void operator()(const char* what) { if (!on) return; auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[corb_lba_store] %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count()); t = n; }
};
struct DevBuf {                       // device memory of one call: bumped out of an arena when the caller lends one (the local BA), else hipMalloc'ed (a config-5 global BA
                                      // needs ~1 GB: not taken from the per-device arena, which never shrinks)
    std::vector<void*> ptrs;
    char* arena = nullptr; size_t cap = 0, used = 0, asked = 0;
    ~DevBuf() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** out, size_t n) {
        const size_t bytes = (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
        asked += bytes;
        if (used + bytes <= cap) { *out = (T*)(arena + used); used += bytes; return hipSuccess; }
        void* p = nullptr; hipError_t e = hipMalloc(&p, bytes); if (e == hipSuccess) { ptrs.push_back(p); *out = (T*)p; } return e;
    }
};
struct HostStage {                    // page-locked host memory of one call, same scheme (pageable std::vector beyond the arena)
    char* arena = nullptr; size_t cap = 0, used = 0, asked = 0;
    std::vector<std::vector<char>> spill;
    template <class T> T* take(size_t n) {
        const size_t bytes = (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
        asked += bytes;
        if (used + bytes <= cap) { T* p = (T*)(arena + used); used += bytes; return p; }
        spill.emplace_back(bytes); return (T*)spill.back().data();
    }
};
}

// the graph of the keyframe slots / map-point slots as device arrays (vertices, per-point edge counts and offsets, edges); `who` names the caller in messages
static int graph_status(const char* who, int status, int n_edges)
{
    if (status & BAS_DUPLICATE_KF) { corb_set_error("%s: a keyframe id occurs twice among the keyframe slots", who); return CORB_ERR_ARG; }
    if (status & BAS_BAD_FEATURE) { corb_set_error("%s: an observation refers to a feature its keyframe does not have", who); return CORB_ERR_ARG; }
    if (n_edges < 0) { corb_set_error("%s: more than 2^31 observations", who); return CORB_ERR_ARG; }
    return CORB_OK;
}
// edge_bound > 0 (local windows): the edge array is sized by the bound (a point has at most max_obs observations), the edges are filled behind the scan without a
// look at their number, and *n_edges_out = -1 -- the caller reads the count and the status word with its own first read-back (corb_ba_staged_device)
static int build_graph(const char* who, CorbKfStore* kf, const int32_t* kf_slots, int n_local, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp,
                       DevBuf& buf, BAStoreDev& d, int* n_edges_out, hipStream_t s, HostStage* hs = nullptr, size_t edge_bound = 0)
{
    memset(&d, 0, sizeof(d));
    d.n_kf = n_kf; d.n_mp = n_mp; d.n_local = n_local; d.max_features = kf->F; d.max_obs = mp->O;
    d.kf_base = kf->base; d.kf_bytes = kf->L.bytes; d.mp_base = mp->base; d.mp_bytes = mp->L.bytes;
    int *dks, *dms;
    HIPCHK(buf.alloc(&dks, (size_t)n_kf)); HIPCHK(buf.alloc(&dms, (size_t)n_mp));
    if (hs) {                                                     // through page-locked staging: the uploads are enqueued, not waited for
        int* a = hs->take<int>((size_t)n_kf); int* b = hs->take<int>((size_t)n_mp);
        if (n_kf) memcpy(a, kf_slots, sizeof(int) * (size_t)n_kf);
        if (n_mp) memcpy(b, mp_slots, sizeof(int) * (size_t)n_mp);
        kf_slots = a; mp_slots = b;
    }
    if (n_kf) HIPCHK(hipMemcpyAsync(dks, kf_slots, sizeof(int) * (size_t)n_kf, hipMemcpyHostToDevice, s));
    if (n_mp) HIPCHK(hipMemcpyAsync(dms, mp_slots, sizeof(int) * (size_t)n_mp, hipMemcpyHostToDevice, s));
    d.kf_slots = dks; d.mp_slots = dms;
    size_t cap = 64; while (cap < 2 * (size_t)n_kf) cap <<= 1;
    HIPCHK(buf.alloc(&d.tab.keys, cap)); HIPCHK(buf.alloc(&d.tab.vals, cap)); d.tab.mask = (unsigned int)(cap - 1);
    HIPCHK(hipMemsetAsync(d.tab.keys, 0xFF, cap * 8, s));
    HIPCHK(buf.alloc(&d.poses, (size_t)n_kf * 16)); HIPCHK(buf.alloc(&d.intr, (size_t)n_kf * 5)); HIPCHK(buf.alloc(&d.pose_fixed, (size_t)n_kf)); HIPCHK(buf.alloc(&d.kf_bad, (size_t)n_kf));
    HIPCHK(buf.alloc(&d.points, (size_t)n_mp * 3)); HIPCHK(buf.alloc(&d.point_fixed, (size_t)n_mp)); HIPCHK(buf.alloc(&d.mp_bad, (size_t)n_mp));
    HIPCHK(buf.alloc(&d.edge_cnt, (size_t)n_mp + 1)); HIPCHK(buf.alloc(&d.edge_off, (size_t)n_mp + 1)); HIPCHK(buf.alloc(&d.status, 1));
    int* scan_tmp; HIPCHK(buf.alloc(&scan_tmp, corb_scan_scratch_ints((size_t)n_mp)));
    HIPCHK(hipMemsetAsync(d.status, 0, sizeof(int), s));
    // vertices, then the edge count of every map point, their prefix sums, then the edges themselves (grouped by map point, in mObservations order)
    bas_launch_vertices(d, s);
    bas_launch_count(d, s);
    corb_launch_exclusive_scan(d.edge_cnt, d.edge_off, (size_t)n_mp, scan_tmp, s);
    HIPCHK(hipGetLastError());
    if (edge_bound > 0) {
        HIPCHK(buf.alloc(&d.edges, edge_bound));
        bas_launch_fill(d, s);
        HIPCHK(hipGetLastError());
        *n_edges_out = -1;
        return CORB_OK;
    }
    int n_edges = 0, status = 0;
    HIPCHK(hipMemcpyAsync(&n_edges, d.edge_off + n_mp, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&status, d.status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    { const int rc_ = graph_status(who, status, n_edges); if (rc_) return rc_; }
    HIPCHK(buf.alloc(&d.edges, (size_t)n_edges));
    bas_launch_fill(d, s);
    HIPCHK(hipGetLastError());
    if (!hs) HIPCHK(hipStreamSynchronize(s));                     // (the local BA's copies follow on the same stream)
    *n_edges_out = n_edges;
    return CORB_OK;
}

static int check_slots(const char* who, CorbKfStore* kf, const int32_t* kf_slots, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp)
{
    if (!kf || !mp || n_kf < 0 || n_mp < 0 || (n_kf > 0 && !kf_slots) || (n_mp > 0 && !mp_slots)) { corb_set_error("%s: bad argument", who); return CORB_ERR_ARG; }
    if (kf->device != mp->device) { corb_set_error("%s: the stores live on different devices", who); return CORB_ERR_ARG; }
    // range and order of a slot list in one pass; lists of a global BA (5 M map points: 2 ms on one thread) on a few threads
    auto scan = [](const int32_t* sl, int n, int cap, bool& in_range, bool& ascending) {
        auto part = [sl, cap](int b, int e, bool& ok, bool& asc) {
            int lo = 0, hi = 0, desc = 0;                            // branch-free: the loop vectorises
            for (int i = b; i < e; i++) { const int v = sl[i]; lo |= v >> 31; hi |= (cap - 1 - v) >> 31; desc |= (i > 0 && sl[i - 1] >= v) ? 1 : 0; }
            ok = !(lo | hi); asc = !desc;
        };
        const int nt = n >= (1 << 20) ? (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1;
        if (nt <= 1) { part(0, n, in_range, ascending); return; }
        std::vector<char> ok((size_t)nt, 1), asc((size_t)nt, 1); std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&, t] { bool a, b; part((int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt), a, b); ok[t] = a; asc[t] = b; });
        for (auto& x : th) x.join();
        in_range = ascending = true;
        for (int t = 0; t < nt; t++) { in_range = in_range && ok[t]; ascending = ascending && asc[t]; }
    };
    bool kf_ok = true, kf_asc = true, mp_ok = true, mp_asc = true;
    scan(kf_slots, n_kf, kf->capacity, kf_ok, kf_asc); scan(mp_slots, n_mp, mp->capacity, mp_ok, mp_asc);
    if (!kf_ok) { corb_set_error("%s: keyframe slot out of range", who); return CORB_ERR_ARG; }
    if (!mp_ok) { corb_set_error("%s: map-point slot out of range", who); return CORB_ERR_ARG; }
    // a slot named twice would be two vertices writing one record (and, in the local BA's finish kernel, two threads rewriting one observation list): refused
    auto dup = [](const int32_t* sl, int n, int cap, bool ascending) {
        if (ascending) return false;                                  // (the usual case, 5 M slots of a global BA included: no bitmap)
        std::vector<uint64_t> seen(((size_t)cap + 63) / 64, 0);
        for (int i = 0; i < n; i++) { uint64_t& w = seen[(size_t)sl[i] >> 6]; const uint64_t b = 1ull << (sl[i] & 63); if (w & b) return true; w |= b; }
        return false;
    };
    if (dup(kf_slots, n_kf, kf->capacity, kf_asc)) { corb_set_error("%s: a keyframe slot is named twice", who); return CORB_ERR_ARG; }
    if (dup(mp_slots, n_mp, mp->capacity, mp_asc)) { corb_set_error("%s: a map-point slot is named twice", who); return CORB_ERR_ARG; }
    return CORB_OK;
}

extern "C" int corb_ba_solve_store(CorbKfStore* kf, const int32_t* kf_slots, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp,
                                   int iterations, int robust, volatile int* stop_flag, uint64_t loop_kf, CorbBAResult* r, const CorbBAOptions* opt)
{
    Lap lap;
    int rc = check_slots("corb_ba_solve_store", kf, kf_slots, n_kf, mp, mp_slots, n_mp); if (rc) return rc;
    if (!r || iterations < 0) { corb_set_error("corb_ba_solve_store: bad argument"); return CORB_ERR_ARG; }
    rc = corb_select_device(kf->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk_kf(kf->mu); std::lock_guard<std::mutex> lk_mp(mp->mu);
    HIPCHK(hipStreamSynchronize(kf->stream)); HIPCHK(hipStreamSynchronize(mp->stream));
    lap("store: slot checks + locks");
    hipStream_t s = mp->stream;
    DevBuf buf;
    BAStoreDev d; int n_edges = 0;
    rc = build_graph("corb_ba_solve_store", kf, kf_slots, n_kf, n_kf, mp, mp_slots, n_mp, buf, d, &n_edges, s); if (rc) return rc;
    lap("store: graph from records");
    // the solve itself, on the device arrays
    CorbBADeviceProblem dp; memset(&dp, 0, sizeof(dp));
    dp.n_poses = n_kf; dp.n_points = n_mp; dp.n_edges = n_edges;
    dp.poses = d.poses; dp.pose_fixed = d.pose_fixed; dp.points = d.points; dp.point_fixed = d.point_fixed; dp.edges = d.edges; dp.intr = d.intr;
    dp.edge_off = d.edge_off;
    rc = corb_ba_solve_device(&dp, iterations, robust, stop_flag, r, kf->device, opt);      // poses / points updated in place on the device
    if (rc) return rc;
    bas_launch_writeback(d, loop_kf, opt ? opt->scale_factor : 0.f, s);
    HIPCHK(hipGetLastError());
    if (r->poses && n_kf) HIPCHK(hipMemcpyAsync(r->poses, d.poses, sizeof(float) * 16 * (size_t)n_kf, hipMemcpyDeviceToHost, s));
    if (r->points && n_mp) HIPCHK(hipMemcpyAsync(r->points, d.points, sizeof(float) * 3 * (size_t)n_mp, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    lap("store: records updated");
    return CORB_OK;
}

// Optimizer::LocalBundleAdjustment on store records (include/corb_accel.h: corb_local_ba_store).  LocalMapping runs it after every keyframe on a window of
// a few keyframes (C/src/LocalMapping.cc:79): the graph is derived on the device like the global one, the window's problem (a few hundred KB) is handed to the
// staged optimiser (corb_ba_solve_staged: its state and the classification between the two optimize() calls live on the host, as in the host-pointer form),
// and one kernel applies vToErase, the estimates and UpdateNormalAndDepth to the records.
extern "C" int corb_local_ba_store(CorbKfStore* kf, const int32_t* kf_slots, int n_local, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp,
                                   const CorbBAStage* stages, int n_stages, float scale_factor, int apply_erase, volatile int* stop_flag,
                                   CorbBAResult* r, int32_t* erase_pairs, int erase_cap, int* n_erase, const CorbBAOptions* opt)
{
    int rc = check_slots("corb_local_ba_store", kf, kf_slots, n_kf, mp, mp_slots, n_mp); if (rc) return rc;
    if (!r || !stages || n_stages < 1 || n_local < 0 || n_local > n_kf || erase_cap < 0 || (erase_cap > 0 && !erase_pairs) || !(scale_factor > 0.f)) { corb_set_error("corb_local_ba_store: bad argument"); return CORB_ERR_ARG; }
    if (n_erase) *n_erase = 0;
    r->iters_done = 0; r->trials_total = 0;
    if (stop_flag && *stop_flag) return CORB_OK;                  // if(pbStopFlag) if(*pbStopFlag) return; (Optimizer.cc:706-708): nothing is touched
    rc = corb_select_device(kf->device); if (rc) return rc;
    Lap lap;
    std::lock_guard<std::mutex> lk_kf(kf->mu); std::lock_guard<std::mutex> lk_mp(mp->mu);
    HIPCHK(hipStreamSynchronize(kf->stream)); HIPCHK(hipStreamSynchronize(mp->stream));
    hipStream_t s = mp->stream;
    lap("locks + stream syncs");
    // the store's scratch, grown to what the previous calls asked for
    if (mp->lba_dev_want > mp->lba_dev_cap || mp->lba_dev_cap == 0) {
        if (mp->lba_dev) (void)hipFree(mp->lba_dev);
        mp->lba_dev = nullptr; mp->lba_dev_cap = 0;
        // (at least 16 MB / 2 MB: a map's first windows grow from call to call, and a call that outgrows its scratch pays hipMalloc'ed arrays and pageable copies)
        const size_t cap = std::min(std::max(mp->lba_dev_want + mp->lba_dev_want / 2, (size_t)16 << 20), (size_t)256 << 20);
        if (hipMalloc((void**)&mp->lba_dev, cap) == hipSuccess) mp->lba_dev_cap = cap; else { mp->lba_dev = nullptr; (void)hipGetLastError(); }
    }
    if (mp->lba_host_want > mp->lba_host_cap || mp->lba_host_cap == 0) {
        if (mp->lba_host) (void)hipHostFree(mp->lba_host);
        mp->lba_host = nullptr; mp->lba_host_cap = 0;
        const size_t cap = std::min(std::max(mp->lba_host_want + mp->lba_host_want / 2, (size_t)2 << 20), (size_t)64 << 20);
        if (hipHostMalloc((void**)&mp->lba_host, cap) == hipSuccess) mp->lba_host_cap = cap; else { mp->lba_host = nullptr; (void)hipGetLastError(); }
    }
    DevBuf buf; buf.arena = mp->lba_dev; buf.cap = mp->lba_dev_cap;
    HostStage hs; hs.arena = mp->lba_host; hs.cap = mp->lba_host_cap;
    struct Want { CorbMpStore* m; DevBuf* b; HostStage* h; ~Want() { m->lba_dev_want = std::max(m->lba_dev_want, std::min(b->asked, (size_t)256 << 20)); m->lba_host_want = std::max(m->lba_host_want, std::min(h->asked, (size_t)64 << 20)); } } want{mp, &buf, &hs};
    BAStoreDev d; int n_edges = 0;
    // Round 5: the window's problem stays on the device -- flattened there, the optimize() calls and the classifications between them run where the estimates are
    // (corb_ba_staged_device); what crosses PCIe is a handful of counts and the result block (the vToErase list, the estimates).  The edge array is sized by its
    // bound, so that the graph kernels and the flattening's first kernels run back to back and the number of edges comes down with the flattening's counts.
    const size_t edge_bound = (size_t)n_mp * (size_t)mp->O;
    const bool dev_route = corb_ba_staged_device_wanted(stages, n_stages) && n_kf > 0 && n_mp > 0 && edge_bound > 0 && edge_bound <= ((size_t)1 << 20);
    rc = build_graph("corb_local_ba_store", kf, kf_slots, n_local, n_kf, mp, mp_slots, n_mp, buf, d, &n_edges, s, &hs, dev_route ? edge_bound : 0); if (rc) return rc;
    lap("graph from records");
    if (dev_route) {
        if (!mp->lba_event) HIPCHK(hipEventCreateWithFlags(&mp->lba_event, hipEventDisableTiming));
        HIPCHK(hipEventRecord(mp->lba_event, s));
        uint8_t* d_outl; HIPCHK(buf.alloc(&d_outl, edge_bound));
        CorbBADeviceProblem dp; memset(&dp, 0, sizeof(dp));
        dp.n_poses = n_kf; dp.n_points = n_mp; dp.n_edges = -1;
        dp.poses = d.poses; dp.pose_fixed = d.pose_fixed; dp.points = d.points; dp.point_fixed = d.point_fixed; dp.edges = d.edges; dp.intr = d.intr; dp.edge_off = d.edge_off;
        int applicable = 0, status = 0;
        rc = corb_ba_staged_device(&dp, stages, n_stages, stop_flag, r, d_outl, mp->lba_event, d.status, &n_edges, &status, kf->device, opt, &applicable);
        if (rc) return rc;
        rc = graph_status("corb_local_ba_store", status, n_edges); if (rc) return rc;
        if (applicable) {                                       // (the optimiser's stream has been waited for: estimates and flags are complete)
            lap("staged solve (device)");
            bas_launch_local_finish(d, d_outl, apply_erase, scale_factor, s);
            // the results the caller reads -- the vToErase list, the estimates -- as one block, one copy
            const int pairs_off = 64, poses_off = (pairs_off + 2 * n_edges + 63) & ~63, points_off = (poses_off + 16 * n_kf + 63) & ~63, words = points_off + 3 * n_mp;
            int* d_block; HIPCHK(buf.alloc(&d_block, (size_t)words));
            bas_launch_local_results(d, d_outl, n_edges, d_block, pairs_off, poses_off, points_off, s);
            HIPCHK(hipGetLastError());
            int* block = hs.take<int>((size_t)words);
            HIPCHK(hipMemcpyAsync(block, d_block, sizeof(int) * (size_t)words, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            const int ne = block[0];                                // vToErase as (index into kf_slots, index into mp_slots), in edge order
            if (ne < 0 || ne > n_edges) { corb_set_error("corb_local_ba_store: %d outlier observations of %d", ne, n_edges); return CORB_ERR_HIP; }
            if (erase_cap > 0 && ne > 0) memcpy(erase_pairs, block + pairs_off, sizeof(int32_t) * 2 * (size_t)std::min(ne, erase_cap));
            if (n_erase) *n_erase = ne;
            if (r->poses && n_kf) memcpy(r->poses, block + poses_off, sizeof(float) * 16 * (size_t)n_kf);
            if (r->points && n_mp) memcpy(r->points, block + points_off, sizeof(float) * 3 * (size_t)n_mp);
            lap("records updated");
            return CORB_OK;
        }
    }
    float* poses = hs.take<float>((size_t)n_kf * 16); float* intr = hs.take<float>((size_t)n_kf * 5); float* points = hs.take<float>((size_t)n_mp * 3);
    float* oposes = hs.take<float>((size_t)n_kf * 16); float* opoints = hs.take<float>((size_t)n_mp * 3);
    uint8_t* pose_fixed = hs.take<uint8_t>((size_t)n_kf); uint8_t* point_fixed = hs.take<uint8_t>((size_t)n_mp); uint8_t* outl = hs.take<uint8_t>((size_t)n_edges);
    CorbBAEdge* edges = hs.take<CorbBAEdge>((size_t)n_edges);
    memset(outl, 0, (size_t)n_edges);
    if (n_kf) { HIPCHK(hipMemcpyAsync(poses, d.poses, sizeof(float) * 16 * (size_t)n_kf, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(intr, d.intr, sizeof(float) * 5 * (size_t)n_kf, hipMemcpyDeviceToHost, s));
                HIPCHK(hipMemcpyAsync(pose_fixed, d.pose_fixed, (size_t)n_kf, hipMemcpyDeviceToHost, s)); }
    if (n_mp) { HIPCHK(hipMemcpyAsync(points, d.points, sizeof(float) * 3 * (size_t)n_mp, hipMemcpyDeviceToHost, s)); HIPCHK(hipMemcpyAsync(point_fixed, d.point_fixed, (size_t)n_mp, hipMemcpyDeviceToHost, s)); }
    if (n_edges) HIPCHK(hipMemcpyAsync(edges, d.edges, sizeof(CorbBAEdge) * (size_t)n_edges, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    lap("problem to host");
    CorbBAProblem hp; memset(&hp, 0, sizeof(hp));
    hp.n_poses = n_kf; hp.n_points = n_mp; hp.n_edges = n_edges;
    hp.poses = poses; hp.pose_fixed = pose_fixed; hp.points = points; hp.point_fixed = point_fixed; hp.edges = edges; hp.intr = intr;
    float* user_poses = r->poses; float* user_points = r->points;
    r->poses = oposes; r->points = opoints;
    rc = corb_ba_solve_staged(&hp, stages, n_stages, stop_flag, r, outl, kf->device, opt);
    r->poses = user_poses; r->points = user_points;
    if (rc) return rc;
    lap("staged solve");
    uint8_t* d_outl; HIPCHK(buf.alloc(&d_outl, (size_t)n_edges));
    if (n_kf) HIPCHK(hipMemcpyAsync(d.poses, oposes, sizeof(float) * 16 * (size_t)n_kf, hipMemcpyHostToDevice, s));
    if (n_mp) HIPCHK(hipMemcpyAsync(d.points, opoints, sizeof(float) * 3 * (size_t)n_mp, hipMemcpyHostToDevice, s));
    if (n_edges) HIPCHK(hipMemcpyAsync(d_outl, outl, (size_t)n_edges, hipMemcpyHostToDevice, s));
    bas_launch_local_finish(d, d_outl, apply_erase, scale_factor, s);
    HIPCHK(hipGetLastError());
    int ne = 0;                                                   // vToErase as (index into kf_slots, index into mp_slots), in edge order
    for (int e = 0; e < n_edges; e++) if (outl[e]) { if (ne < erase_cap) { erase_pairs[2 * (size_t)ne] = edges[e].pose; erase_pairs[2 * (size_t)ne + 1] = edges[e].point; } ne++; }
    if (n_erase) *n_erase = ne;
    if (user_poses && n_kf) memcpy(user_poses, oposes, sizeof(float) * 16 * (size_t)n_kf);
    if (user_points && n_mp) memcpy(user_points, opoints, sizeof(float) * 3 * (size_t)n_mp);
    HIPCHK(hipStreamSynchronize(s));
    lap("records updated");
    return CORB_OK;
}
