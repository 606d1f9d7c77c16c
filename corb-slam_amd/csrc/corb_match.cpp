// corb_match.cpp -- C-ABI host side of the descriptor matchers (see include/corb_accel.h).
// Flattens the reference's merge-walk over two DBoW2::FeatureVectors (corbslam_client/src/ORBmatcher.cc:
// 183-270, 686-769, 822-921) into a list of common vocabulary nodes, ships the flat arrays to the
// device and launches the wavefront matchers of match_kernels.hip.  No CPU compute fallback: all
// Hamming distances, ratio tests, epipolar tests and the rotation-histogram filter run on the GPU.
#include "corb_internal.h"
#include "match_internal.h"
#include "corb_workspace.h"
#include <vector>
#include <algorithm>
#include <cstring>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

namespace {
struct DevArena {          // one allocation, bump-carved, freed on scope exit
    char* base = nullptr; size_t size = 0, used = 0;
    std::vector<std::pair<size_t, std::pair<const void*, size_t>>> uploads;
    CorbScratch scratch;                               // per-device workspace: no hipMalloc / hipFree per call
    size_t reserve(size_t bytes) { size_t off = (used + 255) & ~(size_t)255; used = off + bytes; return off; }
};
}

extern "C" int corb_descriptor_distance(const uint8_t* a, const uint8_t* b, int n, int32_t* dist, int device)
{
    if (n < 0 || (n > 0 && (!a || !b || !dist))) return CORB_ERR_ARG;
    if (n == 0) return CORB_OK;
    int rc = corb_select_device(device); if (rc) return rc;
    CorbScratch scratch;
    uint8_t* d = nullptr;
    const size_t nb = (size_t)n * 32;
    HIPCHK(scratch.alloc(&d, 2 * nb + (size_t)n * 4 + 512));
    uint8_t* da = d, * db = d + ((nb + 255) & ~(size_t)255); int* dd = (int*)(db + ((nb + 255) & ~(size_t)255));
    hipError_t e = hipMemcpy(da, a, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b, nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) { corb_launch_hamming_pairs(da, db, n, dd, scratch.stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(scratch.stream);
    if (e == hipSuccess) e = hipMemcpy(dist, dd, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { corb_set_error("corb_descriptor_distance: %s", hipGetErrorString(e)); return CORB_ERR_HIP; }
    return CORB_OK;
}

// merge-walk of two ascending node-id lists -> common nodes (ORBmatcher.cc:183-270)
static void common_nodes(const CorbFeatVec& f1, const CorbFeatVec& f2, std::vector<int>& pa, std::vector<int>& pb)
{
    int a = 0, b = 0;
    while (a < f1.n_nodes && b < f2.n_nodes) {
        if (f1.node_id[a] == f2.node_id[b]) { pa.push_back(a); pb.push_back(b); a++; b++; }
        else if (f1.node_id[a] < f2.node_id[b]) a++;       // lower_bound
        else b++;
    }
}

static bool featvec_ok(const CorbFeatVec& f, int n)
{
    if (f.n_nodes < 0) return false;
    if (f.n_nodes == 0) return true;
    if (!f.node_id || !f.offset) return false;
    if (f.offset[0] != 0) return false;
    for (int i = 0; i < f.n_nodes; i++) {
        if (f.offset[i + 1] < f.offset[i]) return false;
        if (f.offset[i + 1] - f.offset[i] > 4096) return false;       // 64 lanes x 64-bit claim mask
        if (i && f.node_id[i] <= f.node_id[i - 1]) return false;
    }
    const int tot = f.offset[f.n_nodes];
    if (tot > 0 && !f.idx) return false;
    for (int i = 0; i < tot; i++) if ((int)f.idx[i] >= n) return false;
    return true;
}

extern "C" int corb_search_by_bow(int variant, const CorbBowSide* A, const CorbBowSide* B, float nnratio,
                                  int check_orientation, int32_t* match, int* n_matches, int device)
{
    if (!A || !B || !match || !n_matches || (variant != 0 && variant != 1)) { corb_set_error("corb_search_by_bow: bad argument"); return CORB_ERR_ARG; }
    const int n1 = A->n, n2 = B->n;
    if (n1 < 0 || n2 < 0 || n1 > 65535 || n2 > 65535 || !featvec_ok(A->fv, n1) || !featvec_ok(B->fv, n2) ||
        (n1 > 0 && (!A->desc || !A->valid)) || (n2 > 0 && !B->desc) || (variant == 1 && n2 > 0 && !B->valid) ||
        (check_orientation && ((n1 > 0 && !A->angle) || (n2 > 0 && !B->angle)))) {
        corb_set_error("corb_search_by_bow: inconsistent inputs"); return CORB_ERR_ARG;
    }
    const int n_slots = variant == 0 ? n2 : n1;
    *n_matches = 0;
    for (int i = 0; i < n_slots; i++) match[i] = -1;
    std::vector<int> pa, pb;
    common_nodes(A->fv, B->fv, pa, pb);
    if (pa.empty() || n_slots == 0) return CORB_OK;
    int rc = corb_select_device(device); if (rc) return rc;
    const int t1 = A->fv.offset[A->fv.n_nodes], t2 = B->fv.offset[B->fv.n_nodes];
    DevArena ar;
    struct Up { size_t off; const void* src; size_t bytes; };
    std::vector<Up> ups;
    auto plan = [&](const void* src, size_t bytes) { size_t off = ar.reserve(bytes ? bytes : 4); ups.push_back({off, src, bytes}); return off; };
    const size_t o_pa = plan(pa.data(), pa.size() * 4), o_pb = plan(pb.data(), pb.size() * 4);
    const size_t o_off1 = plan(A->fv.offset, (size_t)(A->fv.n_nodes + 1) * 4), o_idx1 = plan(A->fv.idx, (size_t)t1 * 4);
    const size_t o_off2 = plan(B->fv.offset, (size_t)(B->fv.n_nodes + 1) * 4), o_idx2 = plan(B->fv.idx, (size_t)t2 * 4);
    const size_t o_d1 = plan(A->desc, (size_t)n1 * 32), o_d2 = plan(B->desc, (size_t)n2 * 32);
    const size_t o_a1 = plan(check_orientation ? A->angle : nullptr, check_orientation ? (size_t)n1 * 4 : 0);
    const size_t o_a2 = plan(check_orientation ? B->angle : nullptr, check_orientation ? (size_t)n2 * 4 : 0);
    const size_t o_v1 = plan(A->valid, (size_t)n1), o_v2 = plan(variant == 1 ? B->valid : nullptr, variant == 1 ? (size_t)n2 : 0);
    const size_t o_match = ar.reserve((size_t)n_slots * 4), o_bin = ar.reserve((size_t)n_slots * 4);
    const size_t o_hist = ar.reserve(CORB_HISTO_LENGTH * 4 + 4);
    HIPCHK(ar.scratch.alloc(&ar.base, ar.used + 256));
    {   // the planned inputs are adjacent in the arena: ONE copy out of a per-thread staging block instead of a dozen small ones
        size_t lo = (size_t)-1, hi = 0;
        for (auto& u : ups) if (u.bytes) { lo = std::min(lo, u.off); hi = std::max(hi, u.off + u.bytes); }
        if (hi > 0) {
            static thread_local std::vector<char> blob;
            blob.resize(hi - lo);
            for (auto& u : ups) if (u.bytes) memcpy(blob.data() + (u.off - lo), u.src, u.bytes);
            HIPCHK(hipMemcpyAsync(ar.base + lo, blob.data(), hi - lo, hipMemcpyHostToDevice, ar.scratch.stream));
        }
    }
    HIPCHK(hipMemsetAsync(ar.base + o_match, 0xFF, (size_t)n_slots * 4, ar.scratch.stream));
    HIPCHK(hipMemsetAsync(ar.base + o_bin, 0xFF, (size_t)n_slots * 4, ar.scratch.stream));
    HIPCHK(hipMemsetAsync(ar.base + o_hist, 0, CORB_HISTO_LENGTH * 4 + 4, ar.scratch.stream));
    CorbBowDev d;
    d.variant = variant; d.check_ori = check_orientation ? 1 : 0; d.n_pairs = (int)pa.size(); d.nnratio = nnratio;
    d.pair_a = (const int*)(ar.base + o_pa); d.pair_b = (const int*)(ar.base + o_pb);
    d.off1 = (const int*)(ar.base + o_off1); d.idx1 = (const int*)(ar.base + o_idx1);
    d.off2 = (const int*)(ar.base + o_off2); d.idx2 = (const int*)(ar.base + o_idx2);
    d.desc1 = (const unsigned long long*)(ar.base + o_d1); d.desc2 = (const unsigned long long*)(ar.base + o_d2);
    d.angle1 = (const float*)(ar.base + o_a1); d.angle2 = (const float*)(ar.base + o_a2);
    d.valid1 = (const uint8_t*)(ar.base + o_v1); d.valid2 = (const uint8_t*)(ar.base + o_v2);
    d.match = (int*)(ar.base + o_match); d.bin = (int*)(ar.base + o_bin);
    d.hist = (int*)(ar.base + o_hist); d.n_matches = d.hist + CORB_HISTO_LENGTH;
    corb_launch_bow(d, n_slots, ar.scratch.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ar.scratch.stream));
    HIPCHK(hipMemcpy(match, d.match, (size_t)n_slots * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(n_matches, d.n_matches, 4, hipMemcpyDeviceToHost));
    return CORB_OK;
}

extern "C" int corb_search_for_triangulation(const CorbTriSide* A, const CorbTriSide* B, const float* F12, float ex, float ey,
                                             const float* scale2, const float* sigma2_2, int nlevels, int only_stereo,
                                             int check_orientation, int32_t* pairs, int* n_matches, int device)
{
    if (!A || !B || !F12 || !scale2 || !sigma2_2 || nlevels < 1 || nlevels > CORB_MAX_LEVELS || !n_matches) { corb_set_error("corb_search_for_triangulation: bad argument"); return CORB_ERR_ARG; }
    const int n1 = A->n, n2 = B->n;
    if (n1 < 0 || n2 < 0 || n1 > 65535 || n2 > 65535 || !featvec_ok(A->fv, n1) || !featvec_ok(B->fv, n2) ||
        (n1 > 0 && (!A->desc || !A->kp || !A->u_right || !A->has_mappoint || !pairs)) ||
        (n2 > 0 && (!B->desc || !B->kp || !B->u_right || !B->has_mappoint))) {
        corb_set_error("corb_search_for_triangulation: inconsistent inputs"); return CORB_ERR_ARG;
    }
    *n_matches = 0;
    std::vector<int> pa, pb;
    common_nodes(A->fv, B->fv, pa, pb);
    // queries: KF1 features without MapPoint (and stereo if required), :836-847
    std::vector<int> q_idx1, q_node2;
    for (size_t k = 0; k < pa.size(); k++)
        for (int i1 = A->fv.offset[pa[k]]; i1 < A->fv.offset[pa[k] + 1]; i1++) {
            const int idx1 = (int)A->fv.idx[i1];
            if (A->has_mappoint[idx1]) continue;
            if (only_stereo && !(A->u_right[idx1] >= 0)) continue;
            q_idx1.push_back(idx1); q_node2.push_back(pb[k]);
        }
    if (q_idx1.empty() || n2 == 0) return CORB_OK;
    int rc = corb_select_device(device); if (rc) return rc;
    const int t2 = B->fv.offset[B->fv.n_nodes];
    DevArena ar;
    struct Up { size_t off; const void* src; size_t bytes; };
    std::vector<Up> ups;
    auto plan = [&](const void* src, size_t bytes) { size_t off = ar.reserve(bytes ? bytes : 4); ups.push_back({off, src, bytes}); return off; };
    const size_t o_q1 = plan(q_idx1.data(), q_idx1.size() * 4), o_q2 = plan(q_node2.data(), q_node2.size() * 4);
    const size_t o_off2 = plan(B->fv.offset, (size_t)(B->fv.n_nodes + 1) * 4), o_idx2 = plan(B->fv.idx, (size_t)t2 * 4);
    const size_t o_d1 = plan(A->desc, (size_t)n1 * 32), o_d2 = plan(B->desc, (size_t)n2 * 32);
    const size_t o_k1 = plan(A->kp, (size_t)n1 * sizeof(CorbKeyPoint)), o_k2 = plan(B->kp, (size_t)n2 * sizeof(CorbKeyPoint));
    const size_t o_u1 = plan(A->u_right, (size_t)n1 * 4), o_u2 = plan(B->u_right, (size_t)n2 * 4);
    const size_t o_m2 = plan(B->has_mappoint, (size_t)n2);
    const size_t o_sc = plan(scale2, (size_t)nlevels * 4), o_sg = plan(sigma2_2, (size_t)nlevels * 4);
    const size_t o_match = ar.reserve((size_t)n1 * 4), o_bin = ar.reserve((size_t)n1 * 4), o_hist = ar.reserve(CORB_HISTO_LENGTH * 4 + 4);
    HIPCHK(ar.scratch.alloc(&ar.base, ar.used + 256));
    {   // the planned inputs are adjacent in the arena: ONE copy out of a per-thread staging block instead of a dozen small ones
        size_t lo = (size_t)-1, hi = 0;
        for (auto& u : ups) if (u.bytes) { lo = std::min(lo, u.off); hi = std::max(hi, u.off + u.bytes); }
        if (hi > 0) {
            static thread_local std::vector<char> blob;
            blob.resize(hi - lo);
            for (auto& u : ups) if (u.bytes) memcpy(blob.data() + (u.off - lo), u.src, u.bytes);
            HIPCHK(hipMemcpyAsync(ar.base + lo, blob.data(), hi - lo, hipMemcpyHostToDevice, ar.scratch.stream));
        }
    }
    HIPCHK(hipMemsetAsync(ar.base + o_match, 0xFF, (size_t)n1 * 4, ar.scratch.stream));
    HIPCHK(hipMemsetAsync(ar.base + o_bin, 0xFF, (size_t)n1 * 4, ar.scratch.stream));
    HIPCHK(hipMemsetAsync(ar.base + o_hist, 0, CORB_HISTO_LENGTH * 4 + 4, ar.scratch.stream));
    CorbTriDev d;
    d.n_queries_dev = nullptr;
    d.n_queries = (int)q_idx1.size(); d.only_stereo = only_stereo ? 1 : 0; d.check_ori = check_orientation ? 1 : 0;
    d.q_idx1 = (const int*)(ar.base + o_q1); d.q_node2 = (const int*)(ar.base + o_q2);
    d.off2 = (const int*)(ar.base + o_off2); d.idx2 = (const int*)(ar.base + o_idx2);
    d.desc1 = (const unsigned long long*)(ar.base + o_d1); d.desc2 = (const unsigned long long*)(ar.base + o_d2);
    d.kp1 = (const CorbKeyPoint*)(ar.base + o_k1); d.kp2 = (const CorbKeyPoint*)(ar.base + o_k2);
    d.uright1 = (const float*)(ar.base + o_u1); d.uright2 = (const float*)(ar.base + o_u2);
    d.has_mp2 = (const uint8_t*)(ar.base + o_m2);
    for (int i = 0; i < 9; i++) d.F12[i] = F12[i];
    d.ex = ex; d.ey = ey;
    d.scale2 = (const float*)(ar.base + o_sc); d.sigma2_2 = (const float*)(ar.base + o_sg);
    d.match = (int*)(ar.base + o_match); d.bin = (int*)(ar.base + o_bin);
    d.hist = (int*)(ar.base + o_hist); d.n_matches = d.hist + CORB_HISTO_LENGTH;
    corb_launch_tri(d, n1, ar.scratch.stream);
    HIPCHK(hipGetLastError());
    std::vector<int> m12(n1);
    HIPCHK(hipStreamSynchronize(ar.scratch.stream));
    HIPCHK(hipMemcpy(m12.data(), d.match, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(n_matches, d.n_matches, 4, hipMemcpyDeviceToHost));
    int k = 0;
    for (int i = 0; i < n1; i++) if (m12[i] >= 0) { pairs[2 * k] = i; pairs[2 * k + 1] = m12[i]; k++; }    // :947-955
    return CORB_OK;
}
