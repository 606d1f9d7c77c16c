// corb_proj.cpp -- C-ABI host side of the projection-guided matchers (see include/corb_accel.h).
// Ships the flat Frame / MapPoint views to the device and launches proj_kernels.hip; the only host arithmetic is the
// frame-to-frame translation test that selects the level window (ORBmatcher.cc:1480-1491).  No CPU compute fallback.
#include "proj_internal.h"
#include <vector>
#include <cstring>

void corb_set_error(const char* fmt, ...);
int corb_select_device(int device);

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

namespace {
struct Arena {
    char* base = nullptr; size_t used = 0;
    struct Up { size_t off; const void* src; size_t bytes; };
    std::vector<Up> ups;
    ~Arena() { if (base) (void)hipFree(base); }
    size_t reserve(size_t bytes) { size_t off = (used + 255) & ~(size_t)255; used = off + (bytes ? bytes : 4); return off; }
    size_t plan(const void* src, size_t bytes) { size_t off = reserve(bytes); ups.push_back({off, src, bytes}); return off; }
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
    HIPCHK(hipMalloc((void**)&ar.base, ar.used + 256));
    for (auto& u : ar.ups) if (u.bytes) HIPCHK(hipMemcpyAsync(ar.base + u.off, u.src, u.bytes, hipMemcpyHostToDevice, nullptr));
    HIPCHK(hipMemsetAsync(ar.base + o_nm, 0, 8, nullptr));
    CorbProjDev d; memset(&d, 0, sizeof(d));
    d.n = n; d.nq = nq; d.min_x = F->min_x; d.min_y = F->min_y; d.max_x = F->max_x; d.max_y = F->max_y;
    d.winv = (float)PROJ_COLS / (F->max_x - F->min_x);                 // mfGridElementWidthInv (Frame.cc:101)
    d.hinv = (float)PROJ_ROWS / (F->max_y - F->min_y);
    for (int l = 0; l < F->nlevels; l++) d.scale[l] = F->scale[l];
    d.nnratio = nnratio; d.ratio_test = ratio_test; d.check_ori = check_ori;
    d.keys = (const CorbKeyPoint*)(ar.base + o_keys); d.u_right = (const float*)(ar.base + o_ur); d.desc = (const unsigned long long*)(ar.base + o_desc);
    d.claimed = (const unsigned char*)(ar.base + o_cl); d.qdesc = (const unsigned long long*)(ar.base + o_qd);
    d.query = (CorbProjQuery*)(ar.base + o_query); d.feat_cell = (int*)(ar.base + o_fc); d.cell_off = (int*)(ar.base + o_co); d.cell_idx = (int*)(ar.base + o_ci);
    d.cand_key = (unsigned long long*)(ar.base + o_ck); d.cand_oct = (unsigned char*)(ar.base + o_oc); d.cand_cnt = (int*)(ar.base + o_cc);
    d.ev_feat = (int*)(ar.base + o_ef); d.ev_bin = (int*)(ar.base + o_eb);
    d.match = (int*)(ar.base + o_match); d.n_matches = (int*)(ar.base + o_nm); d.status = d.n_matches + 1;
    corb_launch_projection(d, mp ? (const CorbTrackedPoint*)(ar.base + o_src) : nullptr, mp ? nullptr : (const CorbLastPoint*)(ar.base + o_src), pose, th, nullptr);
    HIPCHK(hipGetLastError());
    int res[2] = {0, 0};
    HIPCHK(hipMemcpy(res, d.n_matches, 8, hipMemcpyDeviceToHost));
    if (res[1] != 0) { corb_set_error("projection matcher: more than %d candidates in one search window", PROJ_CAND_CAP); return CORB_ERR_OVERFLOW; }
    HIPCHK(hipMemcpy(match, d.match, (size_t)n * 4, hipMemcpyDeviceToHost));
    *n_matches = res[0];
    return CORB_OK;
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
