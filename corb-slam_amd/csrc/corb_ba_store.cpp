// corb_ba_store.cpp -- Optimizer::GlobalBundleAdjustemnt on store records (see include/corb_accel.h: corb_ba_solve_store).
// What the server rank runs after a map push and the re-basing (corbslam_server/src/GlobalOptimize.cpp:435-547 -> corbslam_client/src/Optimizer.cc:43-270):
// the graph is derived on the device from the keyframe / map-point records, solved, and the estimates are written back into the records.
#include "store_host.h"
#include "ba_store_internal.h"
#include "ba_device_problem.h"
#include <vector>
#include <cstring>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

namespace {
struct DevBuf {                       // device memory of one call (a config-5 map needs ~1 GB: not taken from the per-device arena, which never shrinks)
    std::vector<void*> ptrs;
    ~DevBuf() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** out, size_t n) { void* p = nullptr; hipError_t e = hipMalloc(&p, (n ? n : 1) * sizeof(T)); if (e == hipSuccess) { ptrs.push_back(p); *out = (T*)p; } return e; }
};
}

extern "C" int corb_ba_solve_store(CorbKfStore* kf, const int32_t* kf_slots, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp,
                                   int iterations, int robust, volatile int* stop_flag, uint64_t loop_kf, CorbBAResult* r, const CorbBAOptions* opt)
{
    if (!kf || !mp || !r || n_kf < 0 || n_mp < 0 || (n_kf > 0 && !kf_slots) || (n_mp > 0 && !mp_slots) || iterations < 0) { corb_set_error("corb_ba_solve_store: bad argument"); return CORB_ERR_ARG; }
    if (kf->device != mp->device) { corb_set_error("corb_ba_solve_store: the stores live on different devices"); return CORB_ERR_ARG; }
    for (int i = 0; i < n_kf; i++) if (kf_slots[i] < 0 || kf_slots[i] >= kf->capacity) { corb_set_error("corb_ba_solve_store: keyframe slot out of range"); return CORB_ERR_ARG; }
    for (int i = 0; i < n_mp; i++) if (mp_slots[i] < 0 || mp_slots[i] >= mp->capacity) { corb_set_error("corb_ba_solve_store: map-point slot out of range"); return CORB_ERR_ARG; }
    int rc = corb_select_device(kf->device); if (rc) return rc;
    std::lock_guard<std::mutex> lk_kf(kf->mu); std::lock_guard<std::mutex> lk_mp(mp->mu);
    HIPCHK(hipStreamSynchronize(kf->stream)); HIPCHK(hipStreamSynchronize(mp->stream));
    hipStream_t s = mp->stream;
    DevBuf buf;
    BAStoreDev d; memset(&d, 0, sizeof(d));
    d.n_kf = n_kf; d.n_mp = n_mp; d.max_features = kf->F; d.max_obs = mp->O;
    d.kf_base = kf->base; d.kf_bytes = kf->L.bytes; d.mp_base = mp->base; d.mp_bytes = mp->L.bytes;
    int *dks, *dms;
    HIPCHK(buf.alloc(&dks, (size_t)n_kf)); HIPCHK(buf.alloc(&dms, (size_t)n_mp));
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
    int n_edges = 0, status = 0;
    HIPCHK(hipMemcpyAsync(&n_edges, d.edge_off + n_mp, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&status, d.status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (status & BAS_DUPLICATE_KF) { corb_set_error("corb_ba_solve_store: a keyframe id occurs twice among the keyframe slots"); return CORB_ERR_ARG; }
    if (status & BAS_BAD_FEATURE) { corb_set_error("corb_ba_solve_store: an observation refers to a feature its keyframe does not have"); return CORB_ERR_ARG; }
    if (n_edges < 0) { corb_set_error("corb_ba_solve_store: more than 2^31 observations"); return CORB_ERR_ARG; }
    HIPCHK(buf.alloc(&d.edges, (size_t)n_edges));
    bas_launch_fill(d, s);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    // the solve itself, on the device arrays
    CorbBADeviceProblem dp; memset(&dp, 0, sizeof(dp));
    dp.n_poses = n_kf; dp.n_points = n_mp; dp.n_edges = n_edges;
    dp.poses = d.poses; dp.pose_fixed = d.pose_fixed; dp.points = d.points; dp.point_fixed = d.point_fixed; dp.edges = d.edges; dp.intr = d.intr;
    dp.edge_off = d.edge_off;
    rc = corb_ba_solve_device(&dp, iterations, robust, stop_flag, r, kf->device, opt);      // poses / points updated in place on the device
    if (rc) return rc;
    bas_launch_writeback(d, loop_kf, s);
    HIPCHK(hipGetLastError());
    if (r->poses && n_kf) HIPCHK(hipMemcpyAsync(r->poses, d.poses, sizeof(float) * 16 * (size_t)n_kf, hipMemcpyDeviceToHost, s));
    if (r->points && n_mp) HIPCHK(hipMemcpyAsync(r->points, d.points, sizeof(float) * 3 * (size_t)n_mp, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CORB_OK;
}
