// corb_orb.cpp -- C-ABI host side of the ORB extractor and the stereo front-end (see include/corb_accel.h).
// Mirrors the constructor arithmetic of ORB_SLAM2::ORBextractor (corbslam_client/src/ORBextractor.cc:410-470)
// and owns device memory, one HIP stream per handle and the launch sequence.  No CPU compute fallback.
#include "corb_internal.h"
#include <cmath>
#include <cfloat>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <map>
#include <vector>

static thread_local char g_err[512] = "";
void corb_set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* corb_last_error(void) { return g_err; }
extern "C" int corb_version(void) { return 100; }
extern "C" int corb_abi_version(void) { return CORB_ABI_VERSION; }
extern "C" int corb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int corb_pinned_alloc(size_t bytes, void** out)
{
    if (!out) return CORB_ERR_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { corb_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(hipGetLastError())); return CORB_ERR_HIP; }
    return CORB_OK;
}
extern "C" int corb_pinned_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? CORB_OK : CORB_ERR_HIP; }

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { corb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return CORB_ERR_HIP; } } while (0)

static inline int cv_round(double v) { return (int)lrint(v); }            // cvRound (round-half-even)
static inline int cv_floor(double v) { int i = cv_round(v); float d = (float)(v - i); return i - (d < 0); }
static inline int cv_ceil(double v) { int i = cv_round(v); float d = (float)(i - v); return i + (d < 0); }
static inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

int corb_select_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { corb_set_error("no HIP device visible"); return CORB_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) { corb_set_error("device %d out of range (%d visible)", device, n); return CORB_ERR_ARG; }
    HIPCHK(hipSetDevice(device));
    static std::mutex mu; static bool inited[64] = {false};
    std::lock_guard<std::mutex> lk(mu);
    if (device < 64 && !inited[device]) {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            corb_set_error("device %d is %s; libcorb_accel is built for gfx950 only", device, prop.gcnArchName);
            return CORB_ERR_NO_DEVICE;
        }
        corb_orb_device_init();
        HIPCHK(hipGetLastError());
        inited[device] = true;
    }
    return CORB_OK;
}

struct CorbOrb {
    CorbOrbConfig cfg;
    CorbOrbParams p;            // host copy
    CorbOrbParams* dp = nullptr;
    hipStream_t stream = nullptr;
    // A run of many images is issued as `parts` part-batches, part 0 on `stream`, part i on side[i-1]; part i starts when part i-1 has
    // launched its FAST kernel (ev_stage), so the parts run half a pipeline apart and the VALU-bound kernels of one meet the
    // latency-bound kernels of the other.  The side streams are joined into `stream` lazily (corb_join), by the next call that touches the
    // results or the inputs -- back-to-back runs keep their phase offset.
    hipStream_t side[CORB_MAX_PARTS - 1] = {};
    hipEvent_t ev_stage[CORB_MAX_PARTS] = {}, ev_done[CORB_MAX_PARTS - 1] = {};
    int parts = 0;                                // 0: two parts (corb_run_parts); CORB_PARTS fixes another count
    int last_np = 0, max_np = 0;                  // parts of the previous split run; most parts (side streams in use) so far
    bool join_pending = false;
    int last_parts_images = 0;        // images of the last split run (its part boundaries follow from this and last_np)
    size_t octree_lds = 0;
    float scale[CORB_MAX_LEVELS], inv_scale[CORB_MAX_LEVELS], sigma2[CORB_MAX_LEVELS], inv_sigma2[CORB_MAX_LEVELS];
    int quota[CORB_MAX_LEVELS];
    int umax[16];
    int last_n_images = 0;
    std::vector<void*> allocs;
    CorbProfiler prof;
    std::mutex stage_mu;        // guards the pinned staging area below
    int* h_status = nullptr;    // pinned
    int* h_count = nullptr;     // pinned
    // pinned staging of ONE image's input and outputs: the single-image operator (corb_orb_extract) and the fetch calls move their
    // data with true asynchronous DMA and one synchronisation instead of several pageable copies
    uint8_t* d_stage = nullptr;                  // device staging of one contiguous input image (re-pitched by orb_ingest_kernel)
    uint8_t* d_stage_batch = nullptr; size_t stage_batch_bytes = 0;   // staging of a whole batch (corb_orb_upload_batch), allocated on first use
    uint8_t* h_img = nullptr; CorbKeyPoint* h_kp = nullptr; uint8_t* h_desc = nullptr; float* h_f32 = nullptr; int* h_misc = nullptr;
    CorbKeyPoint* d_cand_tmp = nullptr; int* d_cand_n = nullptr; int cand_tmp_cap = 0;
};

template <class T> static int dalloc(CorbOrb* h, T** out, size_t n)
{
    void* ptr = nullptr;
    HIPCHK(hipMalloc(&ptr, n * sizeof(T) + 256));
    h->allocs.push_back(ptr);
    *out = (T*)ptr;
    return CORB_OK;
}

// host replica of the resize coefficient tables (cv::resize INTER_LINEAR, OpenCV 2.4.8 imgwarp.cpp)
static void build_resize_tables(int sw, int sh, int dw, int dh, short* tab)
{
    short* xofs = tab, * xa0 = tab + dw, * xa1 = tab + 2 * dw;
    short* ys0 = tab + 3 * dw, * ys1 = ys0 + dh, * yb0 = ys1 + dh, * yb1 = yb0 + dh;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = (short)sx;
        xa0[dx] = sat_short(cv_round((1.f - fx) * 2048));
        xa1[dx] = sat_short(cv_round(fx * 2048));
        if (sx + 1 >= sw) { xa0[dx] = 2048; xa1[dx] = 0; }      // dx >= xmax : D = S[sx]*ONE
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yb0[dy] = sat_short(cv_round((1.f - fy) * 2048));
        yb1[dy] = sat_short(cv_round(fy * 2048));
        ys0[dy] = (short)(sy < 0 ? 0 : (sy < sh ? sy : sh - 1));
        ys1[dy] = (short)(sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1));
    }
}

extern "C" int corb_orb_create(const CorbOrbConfig* cfg, CorbOrb** out)
{
    if (!cfg || !out) { corb_set_error("null argument"); return CORB_ERR_ARG; }
    *out = nullptr;
    if (cfg->nlevels < 1 || cfg->nlevels > CORB_MAX_LEVELS || cfg->nfeatures < 1 || cfg->max_images < 1 ||
        cfg->width < 1 || cfg->height < 1 || cfg->width > 4000 || cfg->height > 4000 || !(cfg->scale_factor > 1.0f) ||
        cfg->min_th_fast < 1 || cfg->ini_th_fast < cfg->min_th_fast || cfg->ini_th_fast > 255) {
        corb_set_error("invalid CorbOrbConfig"); return CORB_ERR_ARG;
    }
    int rc = corb_select_device(cfg->device);
    if (rc != CORB_OK) return rc;
    CorbOrb* h = new CorbOrb();
    h->cfg = *cfg;
    const int nl = cfg->nlevels;
    // ---- ORBextractor::ORBextractor (ORBextractor.cc:415-469) ----
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) { h->scale[i] = h->scale[i - 1] * cfg->scale_factor; h->sigma2[i] = h->scale[i] * h->scale[i]; }
    for (int i = 0; i < nl; i++) { h->inv_scale[i] = 1.0f / h->scale[i]; h->inv_sigma2[i] = 1.0f / h->sigma2[i]; }
    {
        float factor = 1.0f / cfg->scale_factor;
        float nDesired = cfg->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
        int sum = 0;
        for (int level = 0; level < nl - 1; level++) { h->quota[level] = cv_round(nDesired); sum += h->quota[level]; nDesired *= factor; }
        h->quota[nl - 1] = std::max(cfg->nfeatures - sum, 0);
        int v, v0, vmax = cv_floor(CORB_HALF_PATCH * sqrtf(2.f) / 2 + 1), vmin = cv_ceil(CORB_HALF_PATCH * sqrtf(2.f) / 2);
        const double hp2 = CORB_HALF_PATCH * CORB_HALF_PATCH;
        for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round(sqrt(hp2 - v * v));
        for (v = CORB_HALF_PATCH, v0 = 0; v >= vmin; --v) { while (h->umax[v0] == h->umax[v0 + 1]) ++v0; h->umax[v] = v0; ++v0; }
        static const int expect[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
        for (int i = 0; i < 16; i++) if (h->umax[i] != expect[i]) { corb_set_error("umax table mismatch"); delete h; return CORB_ERR_ARG; }
    }
    // ---- static geometry ----
    CorbOrbParams& p = h->p;
    memset(&p, 0, sizeof(p));
    p.nlevels = nl; p.n_images = cfg->max_images; p.ini_th = cfg->ini_th_fast; p.min_th = cfg->min_th_fast;
    size_t arena = 0; int cells = 0, cands = 0, kps = 0, tiles = 0, tab_off = 0, rec_off = 0;
    for (int l = 0; l < nl; l++) {
        CorbLevel& L = p.lv[l];
        L.w = cv_round((float)cfg->width * h->inv_scale[l]);             // :1111-1112
        L.h = cv_round((float)cfg->height * h->inv_scale[l]);
        L.pitch = (L.w + 63) & ~63;
        L.plane_off = (int)arena;
        arena += ((size_t)L.pitch * L.h + 255) & ~(size_t)255;
        L.maxBX = L.w - CORB_EDGE_THRESHOLD + 3; L.maxBY = L.h - CORB_EDGE_THRESHOLD + 3;   // :775-776
        const float width = (float)(L.maxBX - CORB_MIN_BORDER), height = (float)(L.maxBY - CORB_MIN_BORDER);
        L.nCols = (int)(width / 30.f); L.nRows = (int)(height / 30.f);                        // :784-785
        if (L.nCols < 1 || L.nRows < 1) { corb_set_error("level %d (%dx%d) too small for the 30-px FAST grid", l, L.w, L.h); delete h; return CORB_ERR_ARG; }
        L.wCell = (int)ceilf(width / L.nCols); L.hCell = (int)ceilf(height / L.nRows);        // :786-787
        if (L.wCell > 64 || L.hCell > 64) { corb_set_error("FAST cell %dx%d exceeds the LDS tile", L.wCell, L.hCell); delete h; return CORB_ERR_ARG; }
        L.cell_base = cells; cells += L.nCols * L.nRows;
        L.cell_cap = ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);           // bound on strict 8-neighbour maxima
        L.cand_base = cands; L.cand_cap = L.nCols * L.nRows * L.cell_cap; cands += L.cand_cap;
        L.quota = h->quota[l];
        int nIni = (int)roundf(width / height);                            // :543
        if (nIni < 1) nIni = 1;                                            // reference divides by zero; defined as 1
        L.nIni = nIni;
        L.hX = width / nIni;                                               // :545
        L.node_cap = ((std::max(L.quota + 3, 4 * nIni) + 1) + 3) & ~3;
        L.kp_base = kps; L.kp_cap = L.node_cap; kps += L.kp_cap;
        L.blur_tiles_x = (L.w + 3) / 4; L.blur_tiles_y = (L.h + 31) / 32;     // 4-px column groups x strips of 32 rows (BL_ROWS); CORB_BLUR_T (strip, group) items per workgroup
        L.blur_tile_base = tiles; tiles += (L.blur_tiles_x * L.blur_tiles_y + CORB_BLUR_T - 1) / CORB_BLUR_T;
        L.resize_tab_off = tab_off; tab_off += 3 * L.w + 4 * L.h;
        L.resize_rec_off = rec_off; rec_off += ((L.w + 3) & ~3) + L.h + 4;
        L.scale = h->scale[l];
        L.patch_size = (int)(CORB_PATCH_SIZE * h->scale[l]);              // :835
        p.node_cap_max = std::max(p.node_cap_max, L.node_cap);
        p.ncell_max = std::max(p.ncell_max, L.nCols * L.nRows);
        p.fast_tp = std::max(p.fast_tp, 4 * ((L.wCell + 3) / 4) + 8);     // pad + halo (4) + 4-px groups + right window dword
        p.fast_th = std::max(p.fast_th, L.hCell + 6);
    }
    p.cells_per_image = cells; p.cand_per_image = cands; p.kp_per_image = kps; p.out_cap = kps;
    p.blur_tiles_per_image = tiles; p.arena_per_image = arena;
    h->octree_lds = corb_octree_lds_bytes(p.node_cap_max, p.ncell_max);
    if (h->octree_lds > 160 * 1024) { corb_set_error("quadtree needs %zu B of LDS (> 160 KiB): nfeatures too large", h->octree_lds); delete h; return CORB_ERR_ARG; }
    if (p.node_cap_max > 65535) { corb_set_error("nfeatures too large"); delete h; return CORB_ERR_ARG; }
    // ---- device memory ----
    const size_t NI = (size_t)cfg->max_images;
    short* d_tab = nullptr;
#define DA(ptr, n) do { rc = dalloc(h, &(ptr), (n)); if (rc != CORB_OK) { corb_orb_destroy(h); return rc; } } while (0)
    DA(p.pyr, NI * arena); DA(p.blur, NI * arena);
    DA(p.cell_count, NI * cells); DA(p.cand, NI * cands); DA(p.keys, NI * cands); DA(p.key_node, NI * cands);
    DA(p.kp, NI * kps); DA(p.kp_count, NI * CORB_MAX_LEVELS);
    DA(p.out_kp, NI * p.out_cap); DA(p.out_desc, NI * p.out_cap * 32); DA(p.out_count, NI); DA(p.status, NI);
    DA(d_tab, (size_t)tab_off);
    int2* d_rec = nullptr;
    DA(d_rec, (size_t)rec_off + 4);
    DA(h->dp, 1);
#undef DA
    p.resize_tab = d_tab;
    p.resize_rec = d_rec;
    {
        std::vector<short> tab(tab_off);
        for (int l = 1; l < nl; l++) build_resize_tables(p.lv[l - 1].w, p.lv[l - 1].h, p.lv[l].w, p.lv[l].h, tab.data() + p.lv[l].resize_tab_off);
        std::vector<int2> rec((size_t)rec_off + 4, make_int2(0, 0));
        for (int l = 1; l < nl; l++) {
            const short* tl = tab.data() + p.lv[l].resize_tab_off; const int w = p.lv[l].w, hh = p.lv[l].h;
            int2* xr = rec.data() + p.lv[l].resize_rec_off; int2* yr = xr + ((w + 3) & ~3);
            for (int x = 0; x < ((w + 3) & ~3); x++) { const int xc = std::min(x, w - 1); xr[x] = make_int2(tl[xc], (int)(unsigned short)tl[w + xc] | ((int)(unsigned short)tl[2 * w + xc] << 16)); }
            const short* ys0 = tl + 3 * w; const short* ys1 = ys0 + hh; const short* yb0 = ys1 + hh; const short* yb1 = yb0 + hh;
            for (int y = 0; y < hh; y++) yr[y] = make_int2((int)(unsigned short)ys0[y] | ((int)(unsigned short)ys1[y] << 16), (int)(unsigned short)yb0[y] | ((int)(unsigned short)yb1[y] << 16));
        }
        if (hipMemcpy(d_rec, rec.data(), rec.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) { corb_set_error("resize record upload failed"); corb_orb_destroy(h); return CORB_ERR_HIP; }
        // fused pyramid: per strip and level the rows to build = own share of the level U rows its next level reads
        const int S = 4;
        p.pyr_strips = (nl > 1 && p.lv[nl - 1].h >= 4 * S) ? S : 0;
        for (int s = 0; s < p.pyr_strips; s++) {
            int lo = 0, hi = 0;
            for (int l = nl - 1; l >= 1; l--) {
                const int blo = (int)((long long)s * p.lv[l].h / S), bhi = (int)((long long)(s + 1) * p.lv[l].h / S);
                if (l == nl - 1) { lo = blo; hi = bhi; }
                else {
                    const short* tl = tab.data() + p.lv[l + 1].resize_tab_off;          // tables of level l+1 index rows of level l
                    const short* ys0 = tl + 3 * p.lv[l + 1].w; const short* ys1 = ys0 + p.lv[l + 1].h;
                    const int clo = ys0[lo], chi = ys1[hi - 1] + 1;
                    lo = std::min(blo, clo); hi = std::max(bhi, chi);
                }
                p.pyr_r0[s][l] = (short)lo; p.pyr_r1[s][l] = (short)hi;
            }
        }
        // ... and per column tile the 4-px column groups: own share of the level U the groups its next level reads (sx .. sx + 1 of its first / last pixel).
        // Tiles overlap by a group or two per level, like the strips overlap by a row or two; both neighbours store identical values there.
        const int C = 4;
        p.pyr_ctiles = (p.pyr_strips > 0 && p.lv[nl - 1].w >= 16 * C) ? C : 1;
        for (int c = 0; c < p.pyr_ctiles; c++) {
            int lo = 0, hi = 0;
            for (int l = nl - 1; l >= 1; l--) {
                const int ng = (p.lv[l].w + 3) / 4;
                const int blo = (int)((long long)c * ng / p.pyr_ctiles), bhi = (int)((long long)(c + 1) * ng / p.pyr_ctiles);
                if (l == nl - 1) { lo = blo; hi = bhi; }
                else {
                    const short* xofs = tab.data() + p.lv[l + 1].resize_tab_off;       // source column (level l) of every column of level l+1
                    const int wn = p.lv[l + 1].w, x_first = std::min(4 * lo, wn - 1), x_last = std::min(4 * hi, wn) - 1;
                    const int clo = xofs[x_first] / 4, chi = std::min(xofs[x_last] + 1, p.lv[l].w - 1) / 4 + 1;
                    lo = std::min(blo, clo); hi = std::max(bhi, chi);
                }
                p.pyr_g0[c][l] = (short)lo; p.pyr_g1[c][l] = (short)hi;
            }
        }
        if (hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(short), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->dp, &p, sizeof(p), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(p.status, 0, NI * sizeof(int)) != hipSuccess || hipMemset(p.out_count, 0, NI * sizeof(int)) != hipSuccess ||
            hipMemset(p.pyr, 0, NI * arena) != hipSuccess || hipMemset(p.blur, 0, NI * arena) != hipSuccess ||
            hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
            hipHostMalloc((void**)&h->h_status, NI * sizeof(int)) != hipSuccess ||
            hipHostMalloc((void**)&h->h_count, NI * sizeof(int)) != hipSuccess ||
            hipMalloc((void**)&h->d_stage, (size_t)cfg->width * cfg->height + 256) != hipSuccess ||
            hipHostMalloc((void**)&h->h_img, (size_t)cfg->width * cfg->height) != hipSuccess ||
            hipHostMalloc((void**)&h->h_kp, (size_t)p.out_cap * sizeof(CorbKeyPoint)) != hipSuccess ||
            hipHostMalloc((void**)&h->h_desc, (size_t)p.out_cap * 32) != hipSuccess ||
            hipHostMalloc((void**)&h->h_f32, (size_t)p.out_cap * 2 * sizeof(float)) != hipSuccess ||
            hipHostMalloc((void**)&h->h_misc, 8 * sizeof(int)) != hipSuccess) {
            corb_set_error("device initialisation failed: %s", hipGetErrorString(hipGetLastError()));
            corb_orb_destroy(h); return CORB_ERR_HIP;
        }
    }
    if (const char* e = getenv("CORB_PARTS")) h->parts = std::max(1, std::min(CORB_MAX_PARTS, atoi(e)));
    for (int i = 0; i < CORB_MAX_PARTS; i++)
        if (hipEventCreateWithFlags(&h->ev_stage[i], hipEventDisableTiming) != hipSuccess) { corb_set_error("event creation failed: %s", hipGetErrorString(hipGetLastError())); corb_orb_destroy(h); return CORB_ERR_HIP; }
    *out = h;
    return CORB_OK;
}

extern "C" void corb_orb_destroy(CorbOrb* h)
{
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    for (int i = 0; i < CORB_MAX_PARTS - 1; i++) {
        if (h->side[i]) { (void)hipStreamSynchronize(h->side[i]); (void)hipStreamDestroy(h->side[i]); }
        if (h->ev_done[i]) (void)hipEventDestroy(h->ev_done[i]);
    }
    for (int i = 0; i < CORB_MAX_PARTS; i++) if (h->ev_stage[i]) (void)hipEventDestroy(h->ev_stage[i]);
    for (void* ptr : h->allocs) (void)hipFree(ptr);
    if (h->h_status) (void)hipHostFree(h->h_status);
    if (h->h_count) (void)hipHostFree(h->h_count);
    if (h->d_stage) (void)hipFree(h->d_stage);
    if (h->d_stage_batch) (void)hipFree(h->d_stage_batch);
    if (h->h_img) (void)hipHostFree(h->h_img);
    if (h->h_kp) (void)hipHostFree(h->h_kp);
    if (h->h_desc) (void)hipHostFree(h->h_desc);
    if (h->h_f32) (void)hipHostFree(h->h_f32);
    if (h->h_misc) (void)hipHostFree(h->h_misc);
    if (h->d_cand_tmp) (void)hipFree(h->d_cand_tmp);
    if (h->d_cand_n) (void)hipFree(h->d_cand_n);
    delete h;
}

extern "C" int corb_orb_tables(const CorbOrb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                               int32_t* features_per_level, int32_t* umax)
{
    if (!h) return CORB_ERR_ARG;
    for (int i = 0; i < h->cfg.nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->inv_scale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
        if (features_per_level) features_per_level[i] = h->quota[i];
    }
    if (umax) for (int i = 0; i < 16; i++) umax[i] = h->umax[i];
    return CORB_OK;
}

static void corb_join(CorbOrb* h);

extern "C" int corb_orb_upload(CorbOrb* h, int image, const uint8_t* img, int stride)
{
    if (!h || !img || image < 0 || image >= h->cfg.max_images || stride < h->cfg.width) { corb_set_error("corb_orb_upload: bad argument"); return CORB_ERR_ARG; }
    HIPCHK(hipSetDevice(h->cfg.device));
    corb_join(h);
    const CorbLevel& L0 = h->p.lv[0];
    uint8_t* plane = h->p.pyr + (size_t)image * h->p.arena_per_image + L0.plane_off;
    if (stride == h->cfg.width) {                       // contiguous image: one 1-D copy into the staging buffer, rows laid out on the device
        HIPCHK(hipMemcpyAsync(h->d_stage, img, (size_t)h->cfg.width * h->cfg.height, hipMemcpyHostToDevice, h->stream));
        corb_launch_ingest(h->d_stage, h->cfg.width, h->cfg.height, 1, plane, L0.pitch, 0, h->stream);
        HIPCHK(hipGetLastError());
    } else
        HIPCHK(hipMemcpy2DAsync(plane, L0.pitch, img, stride, h->cfg.width, h->cfg.height, hipMemcpyHostToDevice, h->stream));
    return CORB_OK;
}

/* n contiguous, tightly packed images (n x height x width bytes) -> images first .. first+n-1: ONE host-to-device copy (asynchronous DMA
 * when `imgs` is pinned host memory) and one re-pitching kernel */
extern "C" int corb_orb_upload_batch(CorbOrb* h, int first_image, int n_images, const uint8_t* imgs)
{
    if (!h || !imgs || first_image < 0 || n_images < 1 || first_image + n_images > h->cfg.max_images) { corb_set_error("corb_orb_upload_batch: bad argument"); return CORB_ERR_ARG; }
    HIPCHK(hipSetDevice(h->cfg.device));
    corb_join(h);
    const size_t img_bytes = (size_t)h->cfg.width * h->cfg.height, bytes = img_bytes * n_images;
    if (h->stage_batch_bytes < bytes) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_stage_batch) (void)hipFree(h->d_stage_batch);
        h->d_stage_batch = nullptr; h->stage_batch_bytes = 0;
        HIPCHK(hipMalloc((void**)&h->d_stage_batch, bytes + 256));
        h->stage_batch_bytes = bytes;
    }
    const CorbLevel& L0 = h->p.lv[0];
    HIPCHK(hipMemcpyAsync(h->d_stage_batch, imgs, bytes, hipMemcpyHostToDevice, h->stream));
    corb_launch_ingest(h->d_stage_batch, h->cfg.width, h->cfg.height, n_images, h->p.pyr + (size_t)first_image * h->p.arena_per_image + L0.plane_off, L0.pitch,
                       h->p.arena_per_image, h->stream);
    HIPCHK(hipGetLastError());
    return CORB_OK;
}

/* Results of images first .. first+n-1 with one set of asynchronous copies and one synchronisation: the device arrays are strided by
 * `capacity` = corb_orb_capacity(h) entries per image, and so are the host arrays (keypoints[n][capacity], descriptors[n][capacity][32]);
 * counts[i] = valid entries of image first+i.  Pass pinned host memory for true DMA. */
extern "C" int corb_orb_capacity(CorbOrb* h) { return h ? h->p.out_cap : 0; }
extern "C" int corb_orb_fetch_batch(CorbOrb* h, int first_image, int n_images, CorbKeyPoint* keypoints, uint8_t* descriptors, int32_t* counts)
{
    if (!h || first_image < 0 || n_images < 1 || first_image + n_images > h->cfg.max_images || !counts) { corb_set_error("corb_orb_fetch_batch: bad argument"); return CORB_ERR_ARG; }
    HIPCHK(hipSetDevice(h->cfg.device));
    corb_join(h);
    const size_t cap = (size_t)h->p.out_cap;
    HIPCHK(hipMemcpyAsync(counts, h->p.out_count + first_image, (size_t)n_images * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (keypoints) HIPCHK(hipMemcpyAsync(keypoints, h->p.out_kp + (size_t)first_image * cap, (size_t)n_images * cap * sizeof(CorbKeyPoint), hipMemcpyDeviceToHost, h->stream));
    if (descriptors) HIPCHK(hipMemcpyAsync(descriptors, h->p.out_desc + (size_t)first_image * cap * 32, (size_t)n_images * cap * 32, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return CORB_OK;
}

extern "C" int corb_orb_device_image(CorbOrb* h, int image, void** dptr, size_t* pitch)
{
    if (!h || image < 0 || image >= h->cfg.max_images || !dptr || !pitch) return CORB_ERR_ARG;
    *dptr = h->p.pyr + (size_t)image * h->p.arena_per_image + h->p.lv[0].plane_off;
    *pitch = h->p.lv[0].pitch;
    return CORB_OK;
}

// A run of >= CORB_SPLIT_MIN images is issued as part-batches on the handle's stream and its side streams, staggered by the stage events
// (see CorbOrb).  To the caller it is still one asynchronous operation on the handle: every entry point that reads results or rewrites
// inputs joins the side streams first.
#ifndef CORB_SPLIT_MIN
#define CORB_SPLIT_MIN 32
#endif
static void corb_join(CorbOrb* h)
{
    if (!h->join_pending) return;
    for (int i = 0; i < h->max_np - 1; i++) (void)hipStreamWaitEvent(h->stream, h->ev_done[i], 0);
    h->join_pending = false;
}
// side stream i, created on first use: HIP multiplexes streams onto a few hardware queues, and an idle extra stream per handle made two handles' streams
// share queues (the pipelined host-buffer mode of bench.py lost its transfer / compute overlap: 42.9 k -> 27.6 k fps)
static hipStream_t corb_side(CorbOrb* h, int i)
{
    if (!h->side[i]) {
        (void)hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&h->ev_done[i], hipEventDisableTiming);
    }
    return h->side[i];
}
// units [0, n) (images: ipu = 1, or stereo frames: ipu = 2 images per unit) as parts: launch(first_unit, n_units, stream, stage_event) enqueues one part.
// TWO parts at every size (round 4, tools/gpu_step_sweep.sh, profiles/r04_step_sweep.txt; stereo frames per run -> k stereo fps at 2 / 3 / 4 parts): 128 -> 97.1 / 97.3 /
// 87.8; 192 -> 98.5 / 98.8 / -; 256 -> 101.5 / - / 93.9; 384 -> 101.9; 512 -> 103.4 / 100.6 / 97.4; 768 -> 99.3; 1024 -> 97.2 / - / 98.0.  Larger parts have fewer launch tails
// to fill; more than two in flight only divide the wave slots further.  (Round 2 cut runs into parts of ~128 images: its pyramid kernel had 1 024-thread workgroups.)
template <class Launch>
static void corb_run_parts(CorbOrb* h, int n, int ipu, Launch launch)
{
    int np = h->parts > 0 ? h->parts : 2;
    np = std::min(np, n);
    // Back-to-back runs keep their stagger only when they cut the images the same way.  A run with other part boundaries (another n, another part count, or
    // unsplit) would touch images whose previous part is still in flight on a side stream: join first (free when nothing is pending).
    if (h->join_pending && (np <= 1 || h->last_np != np || h->last_parts_images != n * ipu)) corb_join(h);
    if (np <= 1) { launch(0, n, h->stream, (hipEvent_t) nullptr); return; }
    for (int i = 0; i < np; i++) {
        const int u0 = (int)((long long)n * i / np), u1 = (int)((long long)n * (i + 1) / np);
        hipStream_t st = i == 0 ? h->stream : corb_side(h, i - 1);
        if (i > 0) (void)hipStreamWaitEvent(st, h->ev_stage[i - 1], 0);          // inputs ready (part 0 follows the uploads) + half a pipeline behind part i-1
        // ... and the first part of THIS run stays behind the last part's FAST of the PREVIOUS run: without this second half of the handshake the lag of the
        // side stream is only bounded from below -- any disturbance (one profiled step was enough) let it drift to a full period, i.e. both parts in
        // lockstep, and back-to-back runs stayed in that mode: 81.5 k instead of 85 k fps
        else if (h->last_np > 1) (void)hipStreamWaitEvent(st, h->ev_stage[h->last_np - 1], 0);
        launch(u0, u1 - u0, st, h->ev_stage[i]);
        if (i > 0) (void)hipEventRecord(h->ev_done[i - 1], st);
    }
    h->last_np = np; h->last_parts_images = n * ipu; h->max_np = std::max(np, h->max_np);      // (side streams an earlier, larger run used keep their completed ev_done: joining them again is free)
    h->join_pending = true;
}

extern "C" int corb_orb_run(CorbOrb* h, int n_images)
{
    if (!h || n_images < 1 || n_images > h->cfg.max_images) { corb_set_error("corb_orb_run: bad n_images"); return CORB_ERR_ARG; }
    HIPCHK(hipSetDevice(h->cfg.device));
    CorbProfiler* prof = h->prof.enabled ? &h->prof : nullptr;
    if (n_images >= CORB_SPLIT_MIN && !h->prof.serial)
        corb_run_parts(h, n_images, 1, [&](int first, int n, hipStream_t st, hipEvent_t stage) { corb_launch_orb_pipeline(h->p, first, n, h->octree_lds, st, prof, stage); });
    else {
        corb_join(h);                                      // (an unsplit run after a split one: the side streams may still work on these images)
        corb_launch_orb_pipeline(h->p, 0, n_images, h->octree_lds, h->stream, prof);
    }
    HIPCHK(hipGetLastError());
    h->last_n_images = n_images;
    return CORB_OK;
}

extern "C" int corb_orb_sync(CorbOrb* h)
{
    if (!h) return CORB_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    const int n = h->last_n_images > 0 ? h->last_n_images : h->cfg.max_images;
    corb_join(h);
    HIPCHK(hipMemcpyAsync(h->h_status, h->p.status, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++) if (h->h_status[i] != 0) { corb_set_error("image %d: internal buffer overflow (status %d)", i, h->h_status[i]); return CORB_ERR_OVERFLOW; }
    return CORB_OK;
}

// enqueue the download of one image's results into the pinned staging area (count, status, keypoints, descriptors)
static int corb_orb_stage_results(CorbOrb* h, int image)
{
    HIPCHK(hipMemcpyAsync(h->h_misc, h->p.out_count + image, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_misc + 1, h->p.status + image, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_kp, h->p.out_kp + (size_t)image * h->p.out_cap, (size_t)h->p.out_cap * sizeof(CorbKeyPoint), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_desc, h->p.out_desc + (size_t)image * h->p.out_cap * 32, (size_t)h->p.out_cap * 32, hipMemcpyDeviceToHost, h->stream));
    return CORB_OK;
}
static int corb_orb_unstage(CorbOrb* h, CorbKeyPoint* keypoints, uint8_t* descriptors, int cap, int* n)
{
    const int cnt = h->h_misc[0];
    *n = cnt;
    if (cnt > cap) { corb_set_error("corb_orb_fetch: %d keypoints > capacity %d", cnt, cap); return CORB_ERR_CAPACITY; }
    if (keypoints && cnt > 0) memcpy(keypoints, h->h_kp, (size_t)cnt * sizeof(CorbKeyPoint));
    if (descriptors && cnt > 0) memcpy(descriptors, h->h_desc, (size_t)cnt * 32);
    return CORB_OK;
}
extern "C" int corb_orb_fetch(CorbOrb* h, int image, CorbKeyPoint* keypoints, uint8_t* descriptors, int cap, int* n)
{
    if (!h || image < 0 || image >= h->cfg.max_images || !n) return CORB_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    corb_join(h);
    std::lock_guard<std::mutex> lk(h->stage_mu);                       // one staging area per handle
    int rc = corb_orb_stage_results(h, image); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return corb_orb_unstage(h, keypoints, descriptors, cap, n);
}

extern "C" int corb_orb_extract(CorbOrb* h, const uint8_t* img, int width, int height, int stride,
                                CorbKeyPoint* keypoints, uint8_t* descriptors, int cap, int* n)
{
    if (!h || !n) return CORB_ERR_ARG;
    *n = 0;
    if (!img || width == 0 || height == 0) return CORB_OK;               // _image.empty() (ORBextractor.cc:1046-1047)
    if (width != h->cfg.width || height != h->cfg.height) { corb_set_error("image %dx%d does not match the handle (%dx%d)", width, height, h->cfg.width, h->cfg.height); return CORB_ERR_ARG; }
    if (stride < width) { corb_set_error("corb_orb_extract: stride < width"); return CORB_ERR_ARG; }
    HIPCHK(hipSetDevice(h->cfg.device));
    corb_join(h);
    std::lock_guard<std::mutex> lk(h->stage_mu);
    // image -> pinned staging (one host memcpy) -> device (asynchronous DMA); run; results -> pinned staging; ONE synchronisation
    for (int y = 0; y < height; y++) memcpy(h->h_img + (size_t)y * width, img + (size_t)y * stride, (size_t)width);
    const CorbLevel& L0 = h->p.lv[0];
    HIPCHK(hipMemcpyAsync(h->d_stage, h->h_img, (size_t)width * height, hipMemcpyHostToDevice, h->stream));
    corb_launch_ingest(h->d_stage, width, height, 1, h->p.pyr + L0.plane_off, L0.pitch, 0, h->stream);
    int rc = corb_orb_run(h, 1); if (rc) return rc;
    rc = corb_orb_stage_results(h, 0); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->h_misc[1] != 0) { corb_set_error("image 0: internal buffer overflow (status %d)", h->h_misc[1]); return CORB_ERR_OVERFLOW; }
    return corb_orb_unstage(h, keypoints, descriptors, cap, n);
}

extern "C" int corb_orb_pyramid_level(CorbOrb* h, int image, int level, int blurred, uint8_t* dst, size_t dst_bytes, int* width, int* height)
{
    if (!h || image < 0 || image >= h->cfg.max_images || level < 0 || level >= h->cfg.nlevels) return CORB_ERR_ARG;
    const CorbLevel& L = h->p.lv[level];
    if (width) *width = L.w;
    if (height) *height = L.h;
    if (!dst) return CORB_OK;
    if (dst_bytes < (size_t)L.w * L.h) return CORB_ERR_CAPACITY;
    HIPCHK(hipSetDevice(h->cfg.device));
    corb_join(h);
    const uint8_t* src = (blurred ? h->p.blur : h->p.pyr) + (size_t)image * h->p.arena_per_image + L.plane_off;
    HIPCHK(hipMemcpy2DAsync(dst, L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return CORB_OK;
}

extern "C" int corb_orb_fetch_candidates(CorbOrb* h, int image, int level, CorbKeyPoint* out, int cap, int* n)
{
    if (!h || image < 0 || image >= h->cfg.max_images || level < 0 || level >= h->cfg.nlevels || !n) return CORB_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    corb_join(h);
    const int need = h->p.lv[level].cand_cap;
    if (h->cand_tmp_cap < need) {
        if (h->d_cand_tmp) (void)hipFree(h->d_cand_tmp);
        HIPCHK(hipMalloc((void**)&h->d_cand_tmp, (size_t)need * sizeof(CorbKeyPoint)));
        h->cand_tmp_cap = need;
    }
    if (!h->d_cand_n) HIPCHK(hipMalloc((void**)&h->d_cand_n, sizeof(int)));
    corb_launch_candidates(h->dp, image, level, h->d_cand_tmp, need, h->d_cand_n, h->stream);
    int cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, h->d_cand_n, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *n = cnt;
    if (cnt > cap) return CORB_ERR_CAPACITY;
    if (cnt > 0 && out) HIPCHK(hipMemcpy(out, h->d_cand_tmp, (size_t)cnt * sizeof(CorbKeyPoint), hipMemcpyDeviceToHost));
    return CORB_OK;
}

extern "C" int corb_orb_profile(CorbOrb* h, int enable)
{
    if (!h) return CORB_ERR_ARG;
    h->prof.enabled = enable != 0;
    if (enable) h->prof.reserve(64);                     // two part-batches x 8 kernels x 2 events, twice over
    h->prof.serial = enable == 2;                       // 2: no part-batches, every kernel runs (and is timed) alone
    return CORB_OK;
}

extern "C" int corb_orb_profile_read(CorbOrb* h, CorbKernelTime* out, int cap, int* n)
{
    if (!h || !n) return CORB_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device));
    corb_join(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<CorbKernelTime> acc(h->prof.names.size());
    for (size_t i = 0; i < acc.size(); i++) { memset(&acc[i], 0, sizeof(CorbKernelTime)); snprintf(acc[i].name, sizeof(acc[i].name), "%s", h->prof.names[i].c_str()); }
    for (auto& r : h->prof.recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { acc[r.name_id].total_ms += ms; acc[r.name_id].launches++; }
        h->prof.pool.push_back(r.a); h->prof.pool.push_back(r.b);
    }
    h->prof.recs.clear();
    *n = (int)acc.size();
    for (int i = 0; i < (int)acc.size() && i < cap; i++) out[i] = acc[i];
    return CORB_OK;
}

// ================================ stereo front-end ===============================================
struct CorbStereo {
    CorbOrb* orb = nullptr;
    CorbStereoParams s;
    CorbStereoParams* ds = nullptr;
    int max_frames = 0, last_frames = 0;
    // corb_stereo_frames (the per-frame call of a client): result blocks on the device, the captured kernel chain per frame count, stage events
    CorbStereoFrameLayout lay;
    uint8_t* d_result = nullptr;
    std::map<int, hipGraphExec_t> frame_graph;
    const void* graph_stage = nullptr;       // the staging buffer the captured chains read from (corb_orb_upload_batch may replace it: the chains are then captured again)
    hipEvent_t ev_t[4] = {};
};

extern "C" int corb_stereo_create(const CorbStereoConfig* cfg, CorbStereo** out)
{
    if (!cfg || !out || cfg->max_frames < 1 || !(cfg->fx > 0) || !(cfg->bf > 0)) { corb_set_error("invalid CorbStereoConfig"); return CORB_ERR_ARG; }
    *out = nullptr;
    CorbOrbConfig oc = cfg->orb;
    oc.max_images = 2 * cfg->max_frames;
    CorbOrb* orb = nullptr;
    int rc = corb_orb_create(&oc, &orb);
    if (rc != CORB_OK) return rc;
    CorbStereo* h = new CorbStereo();
    h->orb = orb; h->max_frames = cfg->max_frames;
    CorbStereoParams& s = h->s;
    memset(&s, 0, sizeof(s));
    s.n_frames = cfg->max_frames; s.nlevels = oc.nlevels;
    s.bf = cfg->bf;
    s.mb = cfg->bf / cfg->fx;                // Frame::mb = mbf/fx (Frame.cc:114); read uninitialised by the reference, see DESIGN.md
    for (int i = 0; i < oc.nlevels; i++) { s.scale[i] = orb->scale[i]; s.inv_scale[i] = orb->inv_scale[i]; }
    s.rows0 = orb->p.lv[0].h;
    const size_t NF = (size_t)cfg->max_frames, cap = (size_t)orb->p.out_cap;
    // a right keypoint of octave o is registered on at most 2*ceil(2*scale[o]) + 2 rows (Frame.cc:487-497)
    s.row_cap = orb->p.out_cap * (2 * (int)ceilf(2.0f * orb->scale[oc.nlevels - 1]) + 2);
    if ((size_t)(2 * s.rows0 + 2) * sizeof(int) > 60 * 1024) { corb_set_error("image too tall for the stereo row table"); corb_orb_destroy(orb); delete h; return CORB_ERR_ARG; }
    if (dalloc(orb, &s.u_right, NF * cap) || dalloc(orb, &s.depth, NF * cap) || dalloc(orb, &s.sad, NF * cap) ||
        dalloc(orb, &s.n_matched, NF) || dalloc(orb, &h->ds, 1) ||
        dalloc(orb, &s.row_off, NF * (size_t)(s.rows0 + 1)) || dalloc(orb, &s.row_idx, NF * (size_t)s.row_cap) || dalloc(orb, &s.left_range, NF * cap) ||
        hipMemcpy(h->ds, &s, sizeof(s), hipMemcpyHostToDevice) != hipSuccess) {
        corb_orb_destroy(orb); delete h; return CORB_ERR_HIP;
    }
    {
        // one frame's result block (corb_stereo_frames): 64-byte header, then the six sections, each a multiple of 64 bytes
        auto a64 = [](size_t x) { return (int)((x + 63) & ~(size_t)63); };
        CorbStereoFrameLayout& L = h->lay;
        L.capacity = (int)cap;
        int off = 64;
        L.off_kp_left = off; off += a64(cap * sizeof(CorbKeyPoint));
        L.off_kp_right = off; off += a64(cap * sizeof(CorbKeyPoint));
        L.off_desc_left = off; off += a64(cap * 32);
        L.off_desc_right = off; off += a64(cap * 32);
        L.off_u_right = off; off += a64(cap * sizeof(float));
        L.off_depth = off; off += a64(cap * sizeof(float));
        L.frame_bytes = off;
    }
    *out = h;
    return CORB_OK;
}

extern "C" void corb_stereo_destroy(CorbStereo* h)
{
    if (!h) return;
    (void)hipSetDevice(h->orb->cfg.device);
    if (h->orb->stream) (void)hipStreamSynchronize(h->orb->stream);
    for (auto& g : h->frame_graph) if (g.second) (void)hipGraphExecDestroy(g.second);
    for (auto e : h->ev_t) if (e) (void)hipEventDestroy(e);
    corb_orb_destroy(h->orb); delete h;
}
extern "C" CorbOrb* corb_stereo_orb(CorbStereo* h) { return h ? h->orb : nullptr; }

extern "C" int corb_stereo_upload(CorbStereo* h, int frame, const uint8_t* left, const uint8_t* right, int stride)
{
    if (!h || frame < 0 || frame >= h->max_frames) return CORB_ERR_ARG;
    int rc = corb_orb_upload(h->orb, 2 * frame, left, stride); if (rc) return rc;
    return corb_orb_upload(h->orb, 2 * frame + 1, right, stride);
}

extern "C" int corb_stereo_run(CorbStereo* h, int n_frames)
{
    if (!h || n_frames < 1 || n_frames > h->max_frames) return CORB_ERR_ARG;
    CorbOrb* o = h->orb;
    HIPCHK(hipSetDevice(o->cfg.device));
    CorbProfiler* prof = o->prof.enabled ? &o->prof : nullptr;
    auto launch = [&](int first, int n, hipStream_t st, hipEvent_t stage) {
        corb_launch_orb_pipeline(o->p, 2 * first, 2 * n, o->octree_lds, st, prof, stage);
        corb_launch_stereo(o->p, h->s, first, n, st, prof);
    };
    if (2 * n_frames >= CORB_SPLIT_MIN && !o->prof.serial) corb_run_parts(o, n_frames, 2, launch);     // part-batches of whole frames, see corb_orb_run
    else { corb_join(o); launch(0, n_frames, o->stream, nullptr); }
    HIPCHK(hipGetLastError());
    o->last_n_images = 2 * n_frames;
    h->last_frames = n_frames;
    return CORB_OK;
}

/* frames first .. first+n-1 from ONE contiguous block: per frame the left image followed by the right image (2 x height x width bytes) */
extern "C" int corb_stereo_upload_batch(CorbStereo* h, int first_frame, int n_frames, const uint8_t* left_right)
{
    if (!h || first_frame < 0 || n_frames < 1 || first_frame + n_frames > h->max_frames) return CORB_ERR_ARG;
    return corb_orb_upload_batch(h->orb, 2 * first_frame, 2 * n_frames, left_right);
}
/* stereo results of frames first .. first+n-1: u_right / depth are [n][capacity] (capacity = corb_orb_capacity(corb_stereo_orb(h))),
 * counts[n] = left keypoints per frame, n_matched[n]; one synchronisation.  Keypoints / descriptors: corb_orb_fetch_batch on corb_stereo_orb(h)
 * (image 2f = left, 2f+1 = right). */
extern "C" int corb_stereo_fetch_matches_batch(CorbStereo* h, int first_frame, int n_frames, float* u_right, float* depth, int32_t* n_matched)
{
    if (!h || first_frame < 0 || n_frames < 1 || first_frame + n_frames > h->max_frames) return CORB_ERR_ARG;
    CorbOrb* o = h->orb;
    HIPCHK(hipSetDevice(o->cfg.device));
    corb_join(o);
    const size_t cap = (size_t)o->p.out_cap;
    if (u_right) HIPCHK(hipMemcpyAsync(u_right, h->s.u_right + (size_t)first_frame * cap, (size_t)n_frames * cap * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (depth) HIPCHK(hipMemcpyAsync(depth, h->s.depth + (size_t)first_frame * cap, (size_t)n_frames * cap * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    if (n_matched) HIPCHK(hipMemcpyAsync(n_matched, h->s.n_matched + first_frame, (size_t)n_frames * sizeof(int), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
    return CORB_OK;
}

extern "C" int corb_stereo_frame_layout(CorbStereo* h, CorbStereoFrameLayout* out)
{
    if (!h || !out) return CORB_ERR_ARG;
    *out = h->lay;
    return CORB_OK;
}

// Frame::Frame(stereo) as the reference's client calls it -- per frame (C/src/Frame.cc:61-117, Tracking.cc:166-203) -- in one call: see include/corb_accel.h.
// The kernels of the call (image re-pitching, the ORB chain on the 2n images unsplit, the three stereo kernels, the pack kernel: 12 launches) depend on nothing
// but n, so they are captured once per n and replayed as ONE hipGraph launch between the two transfers.
extern "C" int corb_stereo_frames(CorbStereo* h, int n_frames, const uint8_t* images, void* result, CorbStereoFrameTiming* timing)
{
    if (!h || n_frames < 1 || n_frames > h->max_frames || !images || !result) { corb_set_error("corb_stereo_frames: bad argument"); return CORB_ERR_ARG; }
    CorbOrb* o = h->orb;
    HIPCHK(hipSetDevice(o->cfg.device));
    corb_join(o);
    hipStream_t st = o->stream;
    const size_t img_bytes = (size_t)o->cfg.width * o->cfg.height, in_bytes = img_bytes * 2 * n_frames;
    if (o->stage_batch_bytes < in_bytes || !h->d_result || h->graph_stage != o->d_stage_batch) {
        HIPCHK(hipStreamSynchronize(st));
        for (auto& g : h->frame_graph) if (g.second) (void)hipGraphExecDestroy(g.second);      // (the captured chains hold the staging addresses)
        h->frame_graph.clear();
        if (o->stage_batch_bytes < in_bytes) {
            if (o->d_stage_batch) (void)hipFree(o->d_stage_batch);
            o->d_stage_batch = nullptr; o->stage_batch_bytes = 0;
            const size_t want = img_bytes * 2 * std::min(h->max_frames, std::max(n_frames, 8));     // (room for the usual small frame counts: no re-capture when n grows)
            HIPCHK(hipMalloc((void**)&o->d_stage_batch, want + 256));
            o->stage_batch_bytes = want;
        }
        if (!h->d_result) { if (dalloc(o, &h->d_result, (size_t)h->max_frames * h->lay.frame_bytes)) return CORB_ERR_HIP; HIPCHK(hipMemsetAsync(h->d_result, 0, (size_t)h->max_frames * h->lay.frame_bytes, st)); }
        h->graph_stage = o->d_stage_batch;
    }
    if (timing && !h->ev_t[0]) for (auto& e : h->ev_t) HIPCHK(hipEventCreate(&e));
    CorbProfiler* prof = o->prof.enabled ? &o->prof : nullptr;
    auto chain = [&](CorbProfiler* pr) {
        const CorbLevel& L0 = o->p.lv[0];
        corb_launch_ingest(o->d_stage_batch, o->cfg.width, o->cfg.height, 2 * n_frames, o->p.pyr + L0.plane_off, L0.pitch, o->p.arena_per_image, st);
        corb_launch_orb_pipeline(o->p, 0, 2 * n_frames, o->octree_lds, st, pr);
        corb_launch_stereo(o->p, h->s, 0, n_frames, st, pr);
        corb_launch_stereo_pack(o->p, h->s, 0, n_frames, h->d_result, h->lay, st, pr);
    };
    static const bool no_graph = corb_dev_env("CORB_FRAME_NO_GRAPH") != nullptr;      // -DCORB_DEV builds only: the launches one by one (A/B of the capture)
    if (timing) HIPCHK(hipEventRecord(h->ev_t[0], st));
    HIPCHK(hipMemcpyAsync(o->d_stage_batch, images, in_bytes, hipMemcpyHostToDevice, st));
    if (timing) HIPCHK(hipEventRecord(h->ev_t[1], st));
    if (prof || no_graph) chain(prof);
    else {
        hipGraphExec_t& ge = h->frame_graph[n_frames];
        if (!ge) {
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            chain(nullptr);
            HIPCHK(hipStreamEndCapture(st, &graph));
            const hipError_t e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { ge = nullptr; corb_set_error("corb_stereo_frames: hipGraphInstantiate: %s", hipGetErrorString(e)); return CORB_ERR_HIP; }
        }
        HIPCHK(hipGraphLaunch(ge, st));
    }
    HIPCHK(hipGetLastError());
    if (timing) HIPCHK(hipEventRecord(h->ev_t[2], st));
    HIPCHK(hipMemcpyAsync(result, h->d_result, (size_t)n_frames * h->lay.frame_bytes, hipMemcpyDeviceToHost, st));
    if (timing) HIPCHK(hipEventRecord(h->ev_t[3], st));
    HIPCHK(hipStreamSynchronize(st));
    o->last_n_images = 2 * n_frames; h->last_frames = n_frames;
    if (timing) {
        (void)hipEventElapsedTime(&timing->ms_upload, h->ev_t[0], h->ev_t[1]);
        (void)hipEventElapsedTime(&timing->ms_kernels, h->ev_t[1], h->ev_t[2]);
        (void)hipEventElapsedTime(&timing->ms_download, h->ev_t[2], h->ev_t[3]);
    }
    for (int f = 0; f < n_frames; f++) {
        const int32_t* hd = reinterpret_cast<const int32_t*>(static_cast<const uint8_t*>(result) + (size_t)f * h->lay.frame_bytes);
        if (hd[3] != 0) { corb_set_error("frame %d: internal buffer overflow (status %d)", f, hd[3]); return CORB_ERR_OVERFLOW; }
    }
    return CORB_OK;
}

extern "C" int corb_stereo_sync(CorbStereo* h) { return h ? corb_orb_sync(h->orb) : CORB_ERR_ARG; }

extern "C" int corb_stereo_fetch_matches(CorbStereo* h, int frame, float* u_right, float* depth, int cap, int* n, int* n_matched)
{
    if (!h || frame < 0 || frame >= h->max_frames || !n) return CORB_ERR_ARG;
    CorbOrb* o = h->orb;
    HIPCHK(hipSetDevice(o->cfg.device));
    corb_join(o);
    std::lock_guard<std::mutex> lk(o->stage_mu);
    HIPCHK(hipMemcpyAsync(o->h_misc + 2, o->p.out_count + 2 * frame, sizeof(int), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipMemcpyAsync(o->h_misc + 3, h->s.n_matched + frame, sizeof(int), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipMemcpyAsync(o->h_f32, h->s.u_right + (size_t)frame * o->p.out_cap, (size_t)o->p.out_cap * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipMemcpyAsync(o->h_f32 + o->p.out_cap, h->s.depth + (size_t)frame * o->p.out_cap, (size_t)o->p.out_cap * sizeof(float), hipMemcpyDeviceToHost, o->stream));
    HIPCHK(hipStreamSynchronize(o->stream));
    const int cnt = o->h_misc[2];
    *n = cnt; if (n_matched) *n_matched = o->h_misc[3];
    if (cnt > cap) return CORB_ERR_CAPACITY;
    if (u_right && cnt > 0) memcpy(u_right, o->h_f32, (size_t)cnt * sizeof(float));
    if (depth && cnt > 0) memcpy(depth, o->h_f32 + o->p.out_cap, (size_t)cnt * sizeof(float));
    return CORB_OK;
}

// device views of the LEFT image's results of stereo frame `frame` (image slot 2 * frame) -- internal, for the keyframe store
int corb_stereo_device_frame(CorbStereo* h, int frame, CorbStereoDeviceFrame* out)
{
    if (!h || !out || frame < 0 || frame >= h->max_frames) return CORB_ERR_ARG;
    CorbOrb* o = h->orb; const size_t cap = (size_t)o->p.out_cap;
    if (hipSetDevice(o->cfg.device) != hipSuccess) return CORB_ERR_HIP;
    corb_join(o);                                           // the consumer enqueues on o->stream
    out->kp = o->p.out_kp + (size_t)(2 * frame) * cap; out->desc = o->p.out_desc + (size_t)(2 * frame) * cap * 32;
    out->u_right = h->s.u_right + (size_t)frame * cap; out->depth = h->s.depth + (size_t)frame * cap;
    out->count = o->p.out_count + 2 * frame; out->cap = (int)cap; out->stream = o->stream; out->device = o->cfg.device;
    return CORB_OK;
}
