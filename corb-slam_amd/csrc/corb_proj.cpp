// corb_proj.cpp -- C-ABI host side of the projection-guided matchers (see include/corb_accel.h).
// Ships the flat Frame / MapPoint views to the device and launches proj_kernels.hip; the only host arithmetic is the
// frame-to-frame translation test that selects the level window (ORBmatcher.cc:1480-1491).  No CPU compute fallback.
#include "proj_internal.h"
#include "corb_workspace.h"
#include <vector>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <cstring>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

namespace {
struct Arena {
    char* base = nullptr; size_t used = 0;
    struct Up { size_t off; const void* src; size_t bytes; };
    std::vector<Up> ups;
    CorbScratch scratch;                               // device memory comes from the per-device workspace (no hipMalloc / hipFree per call)
    size_t reserve(size_t bytes) { size_t off = (used + 255) & ~(size_t)255; used = off + (bytes ? bytes : 4); return off; }
    size_t plan(const void* src, size_t bytes) { size_t off = reserve(bytes); ups.push_back({off, src, bytes}); return off; }
    // the planned inputs are adjacent in the arena: they travel as ONE copy out of a per-thread staging block (six separate copies of a few tens of KB
    // each cost more than the matcher's kernels)
    hipError_t upload_all() {
        size_t lo = (size_t)-1, hi = 0;
        for (auto& u : ups) if (u.bytes) { lo = std::min(lo, u.off); hi = std::max(hi, u.off + u.bytes); }
        if (hi == 0) return hipSuccess;
        if (char* st = static_cast<char*>(scratch.ws->host_take(hi - lo))) {          // pinned staging: the copy is asynchronous, the kernels are launched behind it
            for (auto& u : ups) if (u.bytes) memcpy(st + (u.off - lo), u.src, u.bytes);
            return hipMemcpyAsync(base + lo, st, hi - lo, hipMemcpyHostToDevice, scratch.stream);
        }
        static thread_local std::vector<char> blob;
        blob.resize(hi - lo);
        for (auto& u : ups) if (u.bytes) memcpy(blob.data() + (u.off - lo), u.src, u.bytes);
        return hipMemcpyAsync(base + lo, blob.data(), hi - lo, hipMemcpyHostToDevice, scratch.stream);
    }
    // two adjacent result regions in one copy
    hipError_t fetch2(size_t off_a, void* a, size_t bytes_a, size_t off_b, void* b, size_t bytes_b) {
        const size_t lo = std::min(off_a, off_b), hi = std::max(off_a + bytes_a, off_b + bytes_b);
        if (char* st = static_cast<char*>(scratch.ws->host_take(hi - lo))) {
            hipError_t e = hipMemcpyAsync(st, base + lo, hi - lo, hipMemcpyDeviceToHost, scratch.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(scratch.stream);
            if (e != hipSuccess) return e;
            memcpy(a, st + (off_a - lo), bytes_a); memcpy(b, st + (off_b - lo), bytes_b);
            return hipSuccess;
        }
        static thread_local std::vector<char> back;
        back.resize(hi - lo);
        hipError_t e = hipStreamSynchronize(scratch.stream);
        if (e == hipSuccess) e = hipMemcpy(back.data(), base + lo, hi - lo, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        memcpy(a, back.data() + (off_a - lo), bytes_a); memcpy(b, back.data() + (off_b - lo), bytes_b);
        return hipSuccess;
    }
};

int run_projection(const CorbFrameView* F, int nq, const void* qdesc, const CorbTrackedPoint* mp, const CorbLastPoint* last,
                   const CorbProjPose* pose, float th, float nnratio, int ratio_test, int check_ori, int32_t* match, int* n_matches, int device)
{
    if (!F || !match || !n_matches || F->n < 0 || nq < 0 || F->nlevels < 1 || F->nlevels > CORB_MAX_LEVELS || !(F->max_x > F->min_x) || !(F->max_y > F->min_y) ||
        (F->n > 0 && (!F->keys_un || !F->u_right || !F->desc || !F->claimed)) || !F->scale || (nq > 0 && !qdesc)) {
        corb_set_error("projection matcher: bad argument"); return CORB_ERR_ARG;
    }
    if (F->n > 6000 || nq > 60000) { corb_set_error("projection matcher: frame too large (%d features)", F->n); return CORB_ERR_ARG; }
    *n_matches = 0;
    for (int i = 0; i < F->n; i++) match[i] = -1;
    if (F->n == 0 || nq == 0) return CORB_OK;
    int rc = corb_select_device(device); if (rc) return rc;
    const int n = F->n;
    Arena ar;
    const size_t o_keys = ar.plan(F->keys_un, (size_t)n * sizeof(CorbKeyPoint)), o_ur = ar.plan(F->u_right, (size_t)n * 4);
    const size_t o_desc = ar.plan(F->desc, (size_t)n * 32), o_cl = ar.plan(F->claimed, (size_t)n), o_qd = ar.plan(qdesc, (size_t)nq * 32);
    const size_t o_src = mp ? ar.plan(mp, (size_t)nq * sizeof(CorbTrackedPoint)) : ar.plan(last, (size_t)nq * sizeof(CorbLastPoint));
    const size_t o_query = ar.reserve((size_t)nq * sizeof(CorbProjQuery)), o_fc = ar.reserve((size_t)n * 4), o_co = ar.reserve((PROJ_CELLS + 1) * 4);
    const size_t o_ci = ar.reserve((size_t)n * 4), o_ck = ar.reserve((size_t)nq * PROJ_CAND_CAP * 8), o_oc = ar.reserve((size_t)nq * PROJ_CAND_CAP);
    const size_t o_cc = ar.reserve((size_t)nq * 4), o_ef = ar.reserve((size_t)nq * 4), o_eb = ar.reserve((size_t)nq * 4);
    const size_t o_match = ar.reserve((size_t)n * 4), o_nm = ar.reserve(8);
    HIPCHK(ar.scratch.alloc(&ar.base, ar.used + 256));
    HIPCHK(ar.upload_all());
    HIPCHK(hipMemsetAsync(ar.base + o_nm, 0, 8, ar.scratch.stream));
    CorbProjDev d; memset(&d, 0, sizeof(d));
    d.n = n; d.nq = nq; d.min_x = F->min_x; d.min_y = F->min_y; d.max_x = F->max_x; d.max_y = F->max_y;
    d.winv = (float)PROJ_COLS / (F->max_x - F->min_x);                 // mfGridElementWidthInv (Frame.cc:101)
    d.hinv = (float)PROJ_ROWS / (F->max_y - F->min_y);
    for (int l = 0; l < F->nlevels; l++) d.scale[l] = F->scale[l];
    d.nnratio = nnratio; d.ratio_test = ratio_test; d.check_ori = check_ori; d.check_uright = 1; d.th_dist = CORB_TH_HIGH;
    d.keys = (const CorbKeyPoint*)(ar.base + o_keys); d.u_right = (const float*)(ar.base + o_ur); d.desc = (const unsigned long long*)(ar.base + o_desc);
    d.claimed = (const unsigned char*)(ar.base + o_cl); d.qdesc = (const unsigned long long*)(ar.base + o_qd);
    d.query = (CorbProjQuery*)(ar.base + o_query); d.feat_cell = (int*)(ar.base + o_fc); d.cell_off = (int*)(ar.base + o_co); d.cell_idx = (int*)(ar.base + o_ci);
    d.cand_key = (unsigned long long*)(ar.base + o_ck); d.cand_oct = (unsigned char*)(ar.base + o_oc); d.cand_cnt = (int*)(ar.base + o_cc);
    d.ev_feat = (int*)(ar.base + o_ef); d.ev_bin = (int*)(ar.base + o_eb);
    d.match = (int*)(ar.base + o_match); d.n_matches = (int*)(ar.base + o_nm); d.status = d.n_matches + 1;
    corb_launch_projection(d, mp ? (const CorbTrackedPoint*)(ar.base + o_src) : nullptr, mp ? nullptr : (const CorbLastPoint*)(ar.base + o_src), pose, th, ar.scratch.stream);
    HIPCHK(hipGetLastError());
    int res[2] = {0, 0};
    std::vector<int32_t> m2((size_t)n);                 // (match is only handed over when the call succeeds)
    HIPCHK(ar.fetch2(o_nm, res, 8, o_match, m2.data(), (size_t)n * 4));
    if (res[1] != 0) { corb_set_error("projection matcher: more than %d candidates in one search window", PROJ_CAND_CAP); return CORB_ERR_OVERFLOW; }
    memcpy(match, m2.data(), (size_t)n * 4);
    *n_matches = res[0];
    return CORB_OK;
}

// keyframe-target matchers: upload the KeyFrame view + MapPoint views, grid, prepare, then either the greedy resolution
// (relocalisation projection) or the independent best candidate per point (Fuse, SearchBySim3)
int run_points(const CorbKeyFrameView* K, const uint8_t* claimed, const CorbMapPointView* pts, const uint8_t* qdesc, int nq, const CorbProjTf& tf,
               int greedy, int check_ori, int th_dist, int chi2_check, int32_t* match, int* n_matches, int32_t* best_idx, int32_t* best_dist, int device)
{
    if (!K || K->n < 0 || nq < 0 || K->nlevels < 1 || K->nlevels > CORB_MAX_LEVELS || !(K->max_x > K->min_x) || !(K->max_y > K->min_y) || !K->scale ||
        (K->n > 0 && (!K->keys_un || !K->u_right || !K->desc)) || (nq > 0 && (!pts || !qdesc)) || (chi2_check && !K->inv_level_sigma2) || (greedy && K->n > 0 && !claimed)) {
        corb_set_error("keyframe projection matcher: bad argument"); return CORB_ERR_ARG;
    }
    if (K->n > 6000 || nq > 60000) { corb_set_error("keyframe projection matcher: too large (%d features, %d points)", K->n, nq); return CORB_ERR_ARG; }
    if (greedy) { *n_matches = 0; for (int i = 0; i < K->n; i++) match[i] = -1; }
    else for (int i = 0; i < nq; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (K->n == 0 || nq == 0) return CORB_OK;
    int rc = corb_select_device(device); if (rc) return rc;
    const int n = K->n;
    Arena ar;
    std::vector<unsigned char> zero_claimed;
    if (!claimed) { zero_claimed.assign(n, 0); claimed = zero_claimed.data(); }
    const size_t o_keys = ar.plan(K->keys_un, (size_t)n * sizeof(CorbKeyPoint)), o_ur = ar.plan(K->u_right, (size_t)n * 4);
    const size_t o_desc = ar.plan(K->desc, (size_t)n * 32), o_cl = ar.plan(claimed, (size_t)n), o_qd = ar.plan(qdesc, (size_t)nq * 32);
    const size_t o_src = ar.plan(pts, (size_t)nq * sizeof(CorbMapPointView));
    const size_t o_query = ar.reserve((size_t)nq * sizeof(CorbProjQuery)), o_fc = ar.reserve((size_t)n * 4), o_co = ar.reserve((PROJ_CELLS + 1) * 4);
    const size_t o_ci = ar.reserve((size_t)n * 4);
    const size_t o_ck = ar.reserve(greedy ? (size_t)nq * PROJ_CAND_CAP * 8 : 8), o_oc = ar.reserve(greedy ? (size_t)nq * PROJ_CAND_CAP : 8);
    const size_t o_cc = ar.reserve((size_t)nq * 4), o_ef = ar.reserve((size_t)nq * 4), o_eb = ar.reserve((size_t)nq * 4);
    const size_t o_match = ar.reserve((size_t)n * 4), o_nm = ar.reserve(8), o_bi = ar.reserve((size_t)nq * 4), o_bd = ar.reserve((size_t)nq * 4);
    HIPCHK(ar.scratch.alloc(&ar.base, ar.used + 256));
    HIPCHK(ar.upload_all());
    HIPCHK(hipMemsetAsync(ar.base + o_nm, 0, 8, ar.scratch.stream));
    CorbProjDev d; memset(&d, 0, sizeof(d));
    d.n = n; d.nq = nq; d.min_x = K->min_x; d.min_y = K->min_y; d.max_x = K->max_x; d.max_y = K->max_y;
    d.winv = (float)PROJ_COLS / (K->max_x - K->min_x);                 // mfGridElementWidthInv (KeyFrame.cc:44-45 <- Frame.cc:101)
    d.hinv = (float)PROJ_ROWS / (K->max_y - K->min_y);
    for (int l = 0; l < K->nlevels; l++) { d.scale[l] = K->scale[l]; d.inv_sigma2[l] = K->inv_level_sigma2 ? K->inv_level_sigma2[l] : 1.0f; }
    d.nnratio = 0.f; d.ratio_test = 0; d.check_ori = check_ori; d.check_uright = 0; d.th_dist = th_dist; d.chi2_check = chi2_check;
    d.keys = (const CorbKeyPoint*)(ar.base + o_keys); d.u_right = (const float*)(ar.base + o_ur); d.desc = (const unsigned long long*)(ar.base + o_desc);
    d.claimed = (const unsigned char*)(ar.base + o_cl); d.qdesc = (const unsigned long long*)(ar.base + o_qd);
    d.query = (CorbProjQuery*)(ar.base + o_query); d.feat_cell = (int*)(ar.base + o_fc); d.cell_off = (int*)(ar.base + o_co); d.cell_idx = (int*)(ar.base + o_ci);
    d.cand_key = (unsigned long long*)(ar.base + o_ck); d.cand_oct = (unsigned char*)(ar.base + o_oc); d.cand_cnt = (int*)(ar.base + o_cc);
    d.ev_feat = (int*)(ar.base + o_ef); d.ev_bin = (int*)(ar.base + o_eb);
    d.match = (int*)(ar.base + o_match); d.n_matches = (int*)(ar.base + o_nm); d.status = d.n_matches + 1;
    d.best_idx = (int*)(ar.base + o_bi); d.best_dist = (int*)(ar.base + o_bd);
    corb_launch_projection_points(d, (const CorbMapPointView*)(ar.base + o_src), tf, greedy, ar.scratch.stream);
    HIPCHK(hipGetLastError());
    if (greedy) {
        int res[2] = {0, 0};
        std::vector<int32_t> m2((size_t)n);
        HIPCHK(ar.fetch2(o_nm, res, 8, o_match, m2.data(), (size_t)n * 4));
        if (res[1] != 0) { corb_set_error("keyframe projection matcher: more than %d candidates in one search window", PROJ_CAND_CAP); return CORB_ERR_OVERFLOW; }
        memcpy(match, m2.data(), (size_t)n * 4);
        *n_matches = res[0];
    } else HIPCHK(ar.fetch2(o_bi, best_idx, (size_t)nq * 4, o_bd, best_dist, (size_t)nq * 4));
    return CORB_OK;
}
void tf_common(CorbProjTf& tf, const CorbKeyFrameView* K, float fx, float fy, float cx, float cy, float th)
{
    memset(&tf, 0, sizeof(tf));
    tf.fx = fx; tf.fy = fy; tf.cx = cx; tf.cy = cy; tf.bf = K->bf; tf.log_scale = K->log_scale_factor; tf.th = th; tf.nlevels = K->nlevels;
}
void set_affine(float* A, const float* T4x4) { for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) A[i * 4 + j] = T4x4[i * 4 + j]; }
// Ow = -Rcw^T * tcw : exact negation of the transposed rotation, then cv::gemm (double accumulation, one rounding)
void camera_centre(const float* Tcw, float* Ow)
{
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)(-Tcw[k * 4 + i]) * (double)Tcw[k * 4 + 3]; Ow[i] = (float)s; }
}
}  // namespace

extern "C" int corb_search_by_projection_map(const CorbFrameView* frame, const CorbTrackedPoint* points, const uint8_t* point_desc, int n_points,
                                             float th, float nnratio, int32_t* match, int* n_matches, int device)
{
    if (n_points > 0 && !points) { corb_set_error("corb_search_by_projection_map: bad argument"); return CORB_ERR_ARG; }
    if (frame) for (int i = 0; i < n_points; i++) if (points[i].valid && (points[i].level < 0 || points[i].level >= frame->nlevels)) { corb_set_error("corb_search_by_projection_map: level out of range"); return CORB_ERR_ARG; }
    return run_projection(frame, n_points, point_desc, points, nullptr, nullptr, th, nnratio, 1, 0, match, n_matches, device);
}

extern "C" int corb_search_by_projection_frame(const CorbFrameView* cur, const float* Tcw, const float* Tlw, float fx, float fy, float cx, float cy,
                                               float bf, float mb, const CorbLastPoint* last, const uint8_t* last_desc, int n_last,
                                               float th, int mono, int check_orientation, int32_t* match, int* n_matches, int device)
{
    if (!Tcw || !Tlw || (n_last > 0 && !last)) { corb_set_error("corb_search_by_projection_frame: bad argument"); return CORB_ERR_ARG; }
    if (cur) for (int i = 0; i < n_last; i++) if (last[i].valid && (last[i].octave < 0 || last[i].octave >= cur->nlevels)) { corb_set_error("corb_search_by_projection_frame: octave out of range"); return CORB_ERR_ARG; }
    CorbProjPose pose;
    memcpy(pose.Tcw, Tcw, 16 * sizeof(float));
    pose.fx = fx; pose.fy = fy; pose.cx = cx; pose.cy = cy; pose.bf = bf;
    // twc = -Rcw^T tcw ; tlc = Rlw twc + tlw  (cv::gemm on CV_32F: double accumulation, one rounding); forward / backward motion test
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)(-Tcw[k * 4 + i]) * (double)Tcw[k * 4 + 3]; twc[i] = (float)s; }
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)Tlw[i * 4 + k] * (double)twc[k]; tlc[i] = (float)(s + (double)Tlw[i * 4 + 3]); }
    pose.forward = (tlc[2] > mb && !mono) ? 1 : 0;
    pose.backward = (-tlc[2] > mb && !mono) ? 1 : 0;
    return run_projection(cur, n_last, last_desc, nullptr, last, &pose, th, 0.f, 0, check_orientation ? 1 : 0, match, n_matches, device);
}

/* int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize) (ORBmatcher.cc:540-655) */
extern "C" int corb_search_for_initialization(const CorbFrameView* f1, const CorbFrameView* f2, float* prev_matched, int window_size, float nnratio, int check_orientation,
                                              int32_t* matches12, int* n_matches, int device)
{
    if (!f1 || !f2 || !matches12 || !n_matches || f1->n < 0 || f2->n < 0 || (f1->n > 0 && (!f1->keys_un || !f1->desc || !prev_matched)) || (f2->n > 0 && (!f2->keys_un || !f2->desc)) ||
        !(f2->max_x > f2->min_x) || !(f2->max_y > f2->min_y) || window_size < 0) {
        corb_set_error("corb_search_for_initialization: bad argument"); return CORB_ERR_ARG;
    }
    if (f2->n > 6000 || f1->n > 8192) { corb_set_error("corb_search_for_initialization: frame too large (%d / %d features)", f1->n, f2->n); return CORB_ERR_ARG; }
    *n_matches = 0;
    for (int i = 0; i < f1->n; i++) matches12[i] = -1;
    if (f1->n == 0 || f2->n == 0) return CORB_OK;
    int rc = corb_select_device(device); if (rc) return rc;
    const int n = f2->n, nq = f1->n;
    const int cap = std::min(n, 2048);                     // candidates kept per window (a 2 x 100 px window of a dense frame holds more than the other matchers' 256)
    Arena ar;
    const size_t o_keys = ar.plan(f2->keys_un, (size_t)n * sizeof(CorbKeyPoint)), o_desc = ar.plan(f2->desc, (size_t)n * 32);
    const size_t o_k1 = ar.plan(f1->keys_un, (size_t)nq * sizeof(CorbKeyPoint)), o_qd = ar.plan(f1->desc, (size_t)nq * 32), o_pm = ar.plan(prev_matched, (size_t)nq * 8);
    const size_t o_query = ar.reserve((size_t)nq * sizeof(CorbProjQuery)), o_fc = ar.reserve((size_t)n * 4), o_co = ar.reserve((PROJ_CELLS + 1) * 4), o_ci = ar.reserve((size_t)n * 4);
    const size_t o_ck = ar.reserve((size_t)nq * cap * 8), o_oc = ar.reserve((size_t)nq * cap), o_cc = ar.reserve((size_t)nq * 4);
    const size_t o_eb = ar.reserve((size_t)nq * 4), o_bi = ar.reserve((size_t)nq * 4), o_nm = ar.reserve(8);
    HIPCHK(ar.scratch.alloc(&ar.base, ar.used + 256));
    HIPCHK(ar.upload_all());
    HIPCHK(hipMemsetAsync(ar.base + o_nm, 0, 8, ar.scratch.stream));
    CorbProjDev d; memset(&d, 0, sizeof(d));
    d.n = n; d.nq = nq; d.min_x = f2->min_x; d.min_y = f2->min_y; d.max_x = f2->max_x; d.max_y = f2->max_y;
    d.winv = (float)PROJ_COLS / (f2->max_x - f2->min_x); d.hinv = (float)PROJ_ROWS / (f2->max_y - f2->min_y);      // mfGridElementWidthInv / HeightInv (Frame.cc:101-102)
    d.nnratio = nnratio; d.check_ori = check_orientation ? 1 : 0; d.check_uright = 0; d.th_dist = CORB_TH_LOW; d.cand_cap = cap;
    d.keys = (const CorbKeyPoint*)(ar.base + o_keys); d.desc = (const unsigned long long*)(ar.base + o_desc); d.qdesc = (const unsigned long long*)(ar.base + o_qd);
    d.query = (CorbProjQuery*)(ar.base + o_query); d.feat_cell = (int*)(ar.base + o_fc); d.cell_off = (int*)(ar.base + o_co); d.cell_idx = (int*)(ar.base + o_ci);
    d.cand_key = (unsigned long long*)(ar.base + o_ck); d.cand_oct = (unsigned char*)(ar.base + o_oc); d.cand_cnt = (int*)(ar.base + o_cc);
    d.ev_bin = (int*)(ar.base + o_eb); d.best_idx = (int*)(ar.base + o_bi); d.n_matches = (int*)(ar.base + o_nm); d.status = d.n_matches + 1;
    corb_launch_search_for_initialization(d, (const CorbKeyPoint*)(ar.base + o_k1), (float*)(ar.base + o_pm), (float)window_size, ar.scratch.stream);
    HIPCHK(hipGetLastError());
    int res[2] = {0, 0};
    std::vector<int32_t> m2((size_t)nq); std::vector<float> pm2((size_t)nq * 2);
    HIPCHK(ar.fetch2(o_nm, res, 8, o_bi, m2.data(), (size_t)nq * 4));
    HIPCHK(ar.fetch2(o_nm, res, 8, o_pm, pm2.data(), (size_t)nq * 8));
    if (res[1] != 0) { corb_set_error("corb_search_for_initialization: more than %d candidates in one search window", cap); return CORB_ERR_OVERFLOW; }
    memcpy(matches12, m2.data(), (size_t)nq * 4); memcpy(prev_matched, pm2.data(), (size_t)nq * 8);
    *n_matches = res[0];
    return CORB_OK;
}

/* SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1616-1744) */
extern "C" int corb_search_by_projection_reloc(const CorbKeyFrameView* cur, const uint8_t* claimed, const float* Tcw, const CorbMapPointView* points,
                                               const uint8_t* point_desc, int n_points, float th, int orb_dist, int check_orientation,
                                               int32_t* match, int* n_matches, int device)
{
    if (!cur || !Tcw || !match || !n_matches) { corb_set_error("corb_search_by_projection_reloc: bad argument"); return CORB_ERR_ARG; }
    CorbProjTf tf; tf_common(tf, cur, cur->fx, cur->fy, cur->cx, cur->cy, th);
    set_affine(tf.A, Tcw); camera_centre(Tcw, tf.Ow);
    tf.reloc = 1; tf.invz_double = 1; tf.lvl_hi = 1;
    return run_points(cur, claimed, points, point_desc, n_points, tf, 1, check_orientation ? 1 : 0, orb_dist, 0, match, n_matches, nullptr, nullptr, device);
}

// decompose Scw (ORBmatcher.cc:434-438 = :1124-1128): Rcw = sRcw / scw, tcw = Scw.col(3) / scw (a division by the double scale = a float multiply by (float)(1/s)), Ow = -Rcw' tcw
static void decompose_scw(const float* T, CorbProjTf& tf)
{
    const double dd = (double)T[0] * T[0] + (double)T[1] * T[1] + (double)T[2] * T[2];
    const float scw = (float)std::sqrt(dd);
    const float inv = (float)(1.0 / (double)scw);
    float M[16];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[i * 4 + j] = T[i * 4 + j] * inv; M[i * 4 + 3] = T[i * 4 + 3] * inv; }
    M[12] = M[13] = M[14] = 0; M[15] = 1;
    set_affine(tf.A, M); camera_centre(M, tf.Ow);
}

/* int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th) (ORBmatcher.cc:425-538):
 * Fuse's gates (depth, IsInImage, distance invariance, viewing angle) with a float 1/z (:466), octaves [level-1, level], no chi2 test, TH_LOW, and the
 * sequential claim of keyframe features (vpMatched[idx] on entry + the commits of earlier points, :510 / :530) resolved exactly by proj_resolve_kernel. */
extern "C" int corb_search_by_projection_scw(const CorbKeyFrameView* kf, const uint8_t* claimed, const float* Scw, const CorbMapPointView* points,
                                             const uint8_t* point_desc, int n_points, float th, int32_t* match, int* n_matches, int device)
{
    if (!kf || !Scw || !match || !n_matches) { corb_set_error("corb_search_by_projection_scw: bad argument"); return CORB_ERR_ARG; }
    CorbProjTf tf; tf_common(tf, kf, kf->fx, kf->fy, kf->cx, kf->cy, th);
    decompose_scw(Scw, tf);
    tf.invz_double = 0; tf.check_normal = 1; tf.lvl_hi = 0;
    return run_points(kf, claimed, points, point_desc, n_points, tf, 1, 0, CORB_TH_LOW, 0, match, n_matches, nullptr, nullptr, device);
}

/* ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (:960-1116, sim3 = 0) and Fuse(KeyFrame*, cv::Mat Scw, ..., vpReplacePoint) (:1118-1241, sim3 = 1) */
extern "C" int corb_fuse(const CorbKeyFrameView* kf, const float* T, const float* Ow, int sim3, const CorbMapPointView* points, const uint8_t* point_desc,
                         int n_points, float th, int32_t* best_idx, int32_t* best_dist, int* n_fused, int device)
{
    if (!kf || !T || (!sim3 && !Ow) || !best_idx || !best_dist || !n_fused) { corb_set_error("corb_fuse: bad argument"); return CORB_ERR_ARG; }
    CorbProjTf tf; tf_common(tf, kf, kf->fx, kf->fy, kf->cx, kf->cy, th);
    if (sim3) { decompose_scw(T, tf); tf.invz_double = 1; } else { set_affine(tf.A, T); tf.Ow[0] = Ow[0]; tf.Ow[1] = Ow[1]; tf.Ow[2] = Ow[2]; tf.invz_double = 0; }
    tf.check_normal = 1; tf.lvl_hi = 0;
    int rc = run_points(kf, nullptr, points, point_desc, n_points, tf, 0, 0, CORB_TH_LOW, sim3 ? 0 : 1, nullptr, nullptr, best_idx, best_dist, device);
    if (rc) return rc;
    int nf = 0; for (int i = 0; i < n_points; i++) nf += best_idx[i] >= 0;
    *n_fused = nf;
    return CORB_OK;
}

/* ORBmatcher::SearchBySim3(KeyFrame*, KeyFrame*, vpMatches12, s12, R12, t12, th) (:1244-1468) */
extern "C" int corb_search_by_sim3(const CorbKeyFrameView* kf1, const CorbKeyFrameView* kf2, const float* T1w, const float* T2w,
                                   const CorbMapPointView* points1, const uint8_t* desc1, const CorbMapPointView* points2, const uint8_t* desc2,
                                   float s12, const float* R12, const float* t12, float th, int32_t* match12, int* n_found, int device)
{
    if (!kf1 || !kf2 || !T1w || !T2w || !R12 || !t12 || !match12 || !n_found) { corb_set_error("corb_search_by_sim3: bad argument"); return CORB_ERR_ARG; }
    // sR12 = s12*R12 ; sR21 = (1.0/s12)*R12.t() ; t21 = -sR21*t12   (:1262-1264)
    float sR12[9], sR21[9], t21[3];
    const float is = (float)(1.0 / (double)s12);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { sR12[i * 3 + j] = R12[i * 3 + j] * s12; sR21[i * 3 + j] = R12[j * 3 + i] * is; }
    for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)(-sR21[i * 3 + k]) * (double)t12[k]; t21[i] = (float)s; }
    const int N1 = kf1->n, N2 = kf2->n;
    std::vector<int32_t> m1(N1 > 0 ? N1 : 1, -1), m2(N2 > 0 ? N2 : 1, -1), bd((N1 > N2 ? N1 : N2) > 0 ? (N1 > N2 ? N1 : N2) : 1);
    auto direction = [&](const CorbKeyFrameView* B, const float* TAw, const float* sR, const float* t, const CorbMapPointView* pts, const uint8_t* desc, int n, int32_t* out) -> int {
        CorbProjTf tf; tf_common(tf, B, kf1->fx, kf1->fy, kf1->cx, kf1->cy, th);        // the intrinsics of both directions are pKF1's (:1247-1250)
        set_affine(tf.A, TAw);
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) tf.B[i * 4 + j] = sR[i * 3 + j]; tf.B[i * 4 + 3] = t[i]; }
        tf.two = 1; tf.invz_double = 1; tf.dist_from_cam = 1; tf.lvl_hi = 0;
        return run_points(B, nullptr, pts, desc, n, tf, 0, 0, CORB_TH_HIGH, 0, nullptr, nullptr, out, bd.data(), device);
    };
    int rc = direction(kf2, T1w, sR21, t21, points1, desc1, N1, m1.data()); if (rc) return rc;
    rc = direction(kf1, T2w, sR12, t12, points2, desc2, N2, m2.data()); if (rc) return rc;
    int nf = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { match12[i1] = idx2; nf++; }
    }
    *n_found = nf;
    return CORB_OK;
}
