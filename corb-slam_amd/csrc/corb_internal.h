// corb_internal.h -- shared between the HIP kernels and the C-ABI host code of libcorb_accel.so.
// gfx950 (MI355X) only.  Compile everything with -ffp-contract=off: float paths are specified as
// non-fused IEEE operations (see DESIGN.md "numerics contract").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/corb_accel.h"

// Development switches that leave work out or change the solver's path (CORB_ORB_SKIP, CORB_BA_PC_PERIOD, CORB_BA_SMALL_EDGES) exist only in a
// build with -DCORB_DEV (make EXTRA=-DCORB_DEV); the shipped library does not read them.
#include <stdlib.h>
#ifdef CORB_DEV
static inline const char* corb_dev_env(const char* name) { return getenv(name); }
#else
static inline const char* corb_dev_env(const char*) { return nullptr; }
#endif

#define CORB_MAX_LEVELS 16
#define CORB_EDGE_THRESHOLD 19
#define CORB_MIN_BORDER 16        // EDGE_THRESHOLD-3 (C/src/ORBextractor.cc:773)
#define CORB_HALF_PATCH 15
#define CORB_PATCH_SIZE 31
#define CORB_TH_HIGH 100          // C/src/ORBmatcher.cc:37
#define CORB_TH_LOW 50            // C/src/ORBmatcher.cc:38
#define CORB_HISTO_LENGTH 30      // C/src/ORBmatcher.cc:39

// Per-level static geometry (computed on the host at create time from the reference's arithmetic,
// C/src/ORBextractor.cc:415-446, 773-787, 1111-1113).
struct CorbLevel {
    int w, h, pitch;              // level image, pitch in bytes (multiple of 64)
    int plane_off;                // byte offset of this level inside one image's pyramid arena
    int nCols, nRows, wCell, hCell;
    int maxBX, maxBY;             // maxBorderX/Y
    int cell_base;                // first cell of this level in the per-image cell table
    int cell_cap;                 // capacity (entries) of one cell = ceil(wCell/2)*ceil(hCell/2)
    int cand_base;                // first entry of this level in the per-image candidate arena
    int cand_cap;                 // nCols*nRows*cell_cap
    int quota;                    // mnFeaturesPerLevel
    int kp_base, kp_cap;          // per-level slice of the per-image keypoint arrays
    int nIni;                     // initial quadtree nodes
    int node_cap;                 // capacity of the quadtree node table
    int blur_tile_base, blur_tiles_x, blur_tiles_y;   // first workgroup of the level; 4-px column groups; 32-row strips
    int resize_tab_off;           // offset (in shorts) of this level's resize tables
    int resize_rec_off;           // offset (in int2) of this level's packed records (fused pyramid kernel)
    float scale;                  // mvScaleFactor[level]
    float hX;                     // (float)(maxBX-minB)/nIni
    int patch_size;               // (int)(31*scale)
};

#define CORB_BLUR_T 64            // threads per workgroup of the blur kernel (independent wavefronts)
#define CORB_MAX_PARTS 4          // part-batches of one run (corb_orb.cpp): 1 + side streams

struct CorbOrbParams {
    int nlevels, n_images;
    int img_base;                 // first image of this launch (a run is issued as part-batches on two streams)
    int ini_th, min_th;
    int cells_per_image, cand_per_image, kp_per_image, out_cap;   // out_cap: capacity of final per-image arrays
    int blur_tiles_per_image;
    int node_cap_max, ncell_max;  // LDS carve sizes of the quadtree kernel
    int fast_tp, fast_th;         // LDS tile pitch / height of the FAST kernel (max cell + 6)
    int pyr_strips;               // horizontal strips per image of the fused pyramid kernel
    short pyr_r0[8][CORB_MAX_LEVELS], pyr_r1[8][CORB_MAX_LEVELS];   // rows [r0,r1) of level l built by strip s
    int pyr_ctiles;               // column tiles per strip (a workgroup = one strip x one column tile)
    short pyr_g0[8][CORB_MAX_LEVELS], pyr_g1[8][CORB_MAX_LEVELS];   // 4-px column groups [g0,g1) of level l built by column tile c
    size_t arena_per_image;       // bytes of one image's pyramid (== blur) arena
    uint8_t* pyr;                 // [n_images][arena_per_image]
    uint8_t* blur;                // same geometry
    int* cell_count;              // [n_images][cells_per_image]
    uint32_t* cand;               // [n_images][cand_per_image]   packed x | y<<12 | score<<24 (coords - 16)
    uint32_t* keys;               // [n_images][cand_per_image]   compacted candidates of each level
    uint16_t* key_node;           // [n_images][cand_per_image]
    uint32_t* kp;                 // [n_images][kp_per_image]     packed absolute level coords + score
    int* kp_count;                // [n_images][CORB_MAX_LEVELS]
    const short* resize_tab;      // per level: xofs[w], xa0[w], xa1[w], yofs[h], yb0[h], yb1[h]
    const int2* resize_rec;       // per level: xrec[align4(w)] {sx, a0|a1<<16}, yrec[h] {ys0|ys1<<16, b0|b1<<16}
    CorbKeyPoint* out_kp;         // [n_images][out_cap]
    uint8_t* out_desc;            // [n_images][out_cap][32]
    int* out_count;               // [n_images]
    int* status;                  // [n_images] sticky error flags
    CorbLevel lv[CORB_MAX_LEVELS];
};

struct CorbStereoParams {
    int n_frames, nlevels;
    int frame_base;               // first frame of this launch
    float bf, mb;                 // Frame::mbf, Frame::mb (= mbf/fx)
    float scale[CORB_MAX_LEVELS], inv_scale[CORB_MAX_LEVELS];
    int rows0;                    // rows of pyramid level 0
    float* u_right;               // [n_frames][out_cap]
    float* depth;                 // [n_frames][out_cap]
    int* sad;                     // [n_frames][out_cap]  SAD distance of accepted matches, -1 otherwise
    int* n_matched;               // [n_frames]
    int* row_off;                 // [n_frames][rows0+1]  CSR row table of the right keypoints
    int2* row_idx;                // [n_frames][row_cap] candidate = {iR | octave << 16, bits of kp.x}: the matcher needs no second lookup
    int row_cap;
    int2* left_range;             // [n_frames][out_cap]  per LEFT keypoint the candidate range {row_off[row], row_off[row + 1]} of its row (empty for rows outside the image)
};

// XCD-aware work mapping.  MI355X has 8 XCDs with private 4 MiB L2s and workgroup b is observed to run on XCD
// b % 8 (MI355X guide; used for speed only, never for correctness).  Grids are (units, images): the linear
// workgroup id is remapped so that all units of one image run on ONE XCD -- an image's pyramid + blurred
// pyramid (2.9 MB at KITTI size) then stays in that XCD's L2 while its cells / keypoints are processed.
#ifdef __HIPCC__
__device__ __forceinline__ void corb_xcd_remap(int& unit, int& img)
{
    const unsigned nx = gridDim.x, nimg = gridDim.y;
    const unsigned B = blockIdx.x + nx * blockIdx.y;
    const unsigned g = B / (8u * nx), r = B - g * 8u * nx;
    const unsigned in_group = min(8u, nimg - 8u * g);          // the last group may hold fewer than 8 images
    img = (int)(8u * g + r % in_group);
    unit = (int)(r / in_group);
}
#endif

// device views of one stereo frame's results (keyframe store: device-to-device hand-over, corb_store.cpp)
struct CorbStereoDeviceFrame { const CorbKeyPoint* kp; const uint8_t* desc; const float* u_right; const float* depth; const int* count; int cap; hipStream_t stream; int device; };
struct CorbStereo;
int corb_stereo_device_frame(CorbStereo* h, int frame, CorbStereoDeviceFrame* out);

// kernel launchers (orb_kernels.hip / match_kernels.hip); all asynchronous on `stream`
struct CorbProfiler;
void corb_orb_device_init();
void corb_launch_ingest(const uint8_t* stage, int w, int h, int n_images, uint8_t* plane, int pitch, size_t image_stride, hipStream_t stream);   // per device: constant tables + kernel attributes
void corb_launch_stereo_pack(const CorbOrbParams& p, const CorbStereoParams& s, int frame_base, int n_frames, uint8_t* out, const CorbStereoFrameLayout& lay, hipStream_t stream, struct CorbProfiler* prof);
void corb_launch_orb_pipeline(const CorbOrbParams& p, int img_base, int n_images, size_t octree_lds, hipStream_t stream, CorbProfiler* prof, hipEvent_t after_fast = nullptr);
void corb_launch_candidates(const CorbOrbParams* dp, int img, int level, CorbKeyPoint* out, int cap, int* n_out, hipStream_t stream);
void corb_launch_stereo(const CorbOrbParams& p, const CorbStereoParams& s, int frame_base, int n_frames, hipStream_t stream, CorbProfiler* prof);
size_t corb_octree_lds_bytes(int node_cap_max, int ncell_max);

// ---- tiny event profiler: one (start, stop) event pair per launch, resolved at read time ----
#ifdef __HIPCC__
#include <hip/hip_ext.h>
#define CORB_LAUNCH(prof, name, kernel, grid, block, lds, stream, ...) do { \
    if ((prof) && (prof)->enabled) { hipEvent_t ea_, eb_; (prof)->pair(name, &ea_, &eb_); hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, ea_, eb_, 0, __VA_ARGS__); } \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__); } while (0)
#endif
#include <vector>
#include <string>
struct CorbProfiler {
    bool enabled = false;
    bool serial = false;          // corb_orb_profile(h, 2): one stream, no part-batch overlap
    struct Rec { int name_id; hipEvent_t a, b; };
    std::vector<std::string> names;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    int name_id(const char* n) { for (size_t i = 0; i < names.size(); i++) if (names[i] == n) return (int)i; names.push_back(n); return (int)names.size() - 1; }
    void reserve(size_t n) { while (pool.size() < n) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; pool.push_back(e); } }      // event creation costs ~10 us: not inside a timed step
    hipEvent_t get() { if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; } hipEvent_t e; (void)hipEventCreate(&e); return e; }
    // the event pair of one launch: handed to hipExtLaunchKernelGGL, which stamps them with the kernel's OWN start / stop times -- no marker packets around the
    // kernel (event records before and after every kernel serialised the two part-batches of the step they were recorded in: -30 % on that step)
    void pair(const char* n, hipEvent_t* a, hipEvent_t* b) { Rec r; r.name_id = name_id(n); r.a = get(); r.b = get(); recs.push_back(r); *a = r.a; *b = r.b; }
    ~CorbProfiler() { for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); } for (auto e : pool) (void)hipEventDestroy(e); }
};
