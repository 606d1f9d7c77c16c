/* corb_accel.h -- C-ABI of the MI355X-native CORB-SLAM hot path (libcorb_accel.so).
 *
 * Plain C: opaque handles, POD structs, pointers + sizes, int status codes.  No exceptions.  An
 * extractor / stereo handle owns its device memory and its HIP streams (a run is issued as two
 * overlapping half-batches); the handle-less calls (matchers, optimisers, map maintenance) draw
 * device memory and stream from a per-device workspace with a short-call lane and a
 * long-optimisation lane.  Every function may be called from any host thread; calls on different
 * handles / lanes run concurrently (the reference calls the left/right extractors from two
 * std::threads, corbslam_client/src/Frame.cc:78-81, and its matchers / optimisers from the
 * Tracking, LocalMapping and LoopClosing threads).
 *
 * Citations are relative to the reference tree (lifunudt/CORB-SLAM):
 *   C/ = corbslam_client/   S/ = corbslam_server/   G/ = corbslam_client/Thirdparty/g2o/g2o/
 *
 * There is NO CPU fallback: every entry point returns CORB_ERR_NO_DEVICE / CORB_ERR_HIP if the
 * gfx950 device or the kernels are unavailable.
 */
#ifndef CORB_ACCEL_H
#define CORB_ACCEL_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORB_OK              0
#define CORB_ERR_ARG        -1   /* bad argument / size mismatch with the handle's configuration */
#define CORB_ERR_CAPACITY   -2   /* caller-provided output capacity too small (nothing written past cap) */
#define CORB_ERR_HIP        -3   /* a HIP runtime call failed; see corb_last_error() */
#define CORB_ERR_NO_DEVICE  -4   /* no gfx950 device visible */
#define CORB_ERR_OVERFLOW   -5   /* internal fixed-capacity buffer overflowed (reported, never silent) */
#define CORB_ERR_NUMERIC    -6   /* BA: non-finite values */

const char* corb_last_error(void);          /* thread-local, static storage */
int corb_device_count(void);
int corb_version(void);                     /* 100*major + minor */
/* Layout version of the structs of this header.  CorbBAOptions, CorbBAResult and the record structs carry no size field: a caller checks ONCE, after loading,
   that the library it got was built from the header it was compiled against -- corb_abi_version() == CORB_ABI_VERSION -- and refuses to go on otherwise
   (host/corb_host.hpp: corb::check_abi(); the Python harness does it in load()).  6: the map-point record carries CorbMapPointScratch behind its observation lists (corb_mp_store_record_bytes grew), corb_release_scratch and the
   last two matchers exist; 5: CorbBAResult gained pcg_residual_max / _last / grad_inf;
   4 was round 4's layout (CorbBAOptions 32 bytes with pc_multilevel / scale_factor, CorbBAResult.pc_levels). */
#define CORB_ABI_VERSION 6
int corb_abi_version(void);
/* Page-locked host memory from the HIP runtime THIS library is linked with: host buffers handed to corb_*_upload_batch / corb_*_fetch_batch travel by
 * asynchronous DMA only if that runtime knows them as pinned (a buffer pinned through another copy of the runtime loaded in the same process -- e.g. the
 * one a Python framework ships -- is treated as pageable: staged, synchronous copies). */
int corb_pinned_alloc(size_t bytes, void** out);
int corb_pinned_free(void* p);
/* Optional, once per process and device at start-up: creates the per-device workspace lanes (stream, events, page-locked scratch).  Rounds 1-2 also
 * pre-loaded rocBLAS / rocSOLVER here; the library links neither any more (the dense solves are csrc/dense_chol.hip). */
int corb_warmup(int device);
/* Gives the device and page-locked memory the library keeps between calls on `device` back to the runtime: the bump arenas of its two workspace lanes (they grow to what
 * the largest call needed -- ~9 GB after a 50 000-keyframe global BA -- and never shrink by themselves) and the staging of large host-array BA calls (~0.8 GB + 64 MB
 * page-locked at that size).  What a call of another thread holds at this moment is skipped.  *bytes_released (optional) = what was freed.  The stores are not touched. */
int corb_release_scratch(int device, uint64_t* bytes_released);

/* 28-byte POD, bit-identical to cv::KeyPoint as filled by the reference
 * (C/src/ORBextractor.cc:837-847, 1094-1101): pt, size, angle, response, octave, class_id */
typedef struct CorbKeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} CorbKeyPoint;

/* ============================ ORB extraction ==============================================
 * Replaces ORB_SLAM2::ORBextractor (C/include/ORBextractor.h:45-114, C/src/ORBextractor.cc). */
typedef struct CorbOrbConfig {
    int32_t nfeatures;      /* ORBextractor.nFeatures   (C/src/Tracking.cc:112) */
    float   scale_factor;   /* ORBextractor.scaleFactor (:113) */
    int32_t nlevels;        /* ORBextractor.nLevels     (:114) */
    int32_t ini_th_fast;    /* ORBextractor.iniThFAST   (:115) */
    int32_t min_th_fast;    /* ORBextractor.minThFAST   (:116) */
    int32_t width, height;  /* image size this handle is built for (pyramid geometry is static) */
    int32_t max_images;     /* images processed per launch (1 = one eye, drop-in for operator()) */
    int32_t device;         /* HIP device ordinal */
} CorbOrbConfig;

typedef struct CorbOrb CorbOrb;

/* ORBextractor::ORBextractor (C/src/ORBextractor.cc:410-470) */
int corb_orb_create(const CorbOrbConfig* cfg, CorbOrb** out);
void corb_orb_destroy(CorbOrb* h);

/* ORBextractor::operator() (C/src/ORBextractor.cc:1043-1105), one host image in, host results out.
 * `mask` of the reference is ignored there and absent here.  Empty image (img==NULL or w*h==0)
 * => *n = 0, CORB_OK (mirrors :1046-1047).  Synchronous. */
int corb_orb_extract(CorbOrb* h, const uint8_t* img, int width, int height, int stride,
                     CorbKeyPoint* keypoints, uint8_t* descriptors /* cap x 32 */, int cap, int* n);

/* Getters GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (C/include/ORBextractor.h:62-82) + mnFeaturesPerLevel + umax.
 * Any pointer may be NULL.  Arrays have nlevels entries (umax: 16). */
int corb_orb_tables(const CorbOrb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* features_per_level, int32_t* umax);

/* mvImagePyramid[level] of image `image` after the last extraction (C/include/ORBextractor.h:85),
 * copied to host with pitch == *width.  dst may be NULL to query the size. blurred!=0 returns the
 * 7x7-Gaussian working copy used for the descriptors (C/src/ORBextractor.cc:1085-1086). */
int corb_orb_pyramid_level(CorbOrb* h, int image, int level, int blurred, uint8_t* dst, size_t dst_bytes,
                           int* width, int* height);

/* Batched, device-resident form of the same operator (the throughput path).
 *   upload : copy one host image into slot `image` (async on the handle's stream; contiguous images take the fast path:
 *            one 1-D copy + a re-pitching kernel -- a strided 2-D copy of a 1241-byte-wide image costs 2.6 ms)
 *   run    : launch the whole pipeline for images [0, n_images) (async).  A run of many images is issued as two part-batches half a pipeline apart (the
 *            second on a side stream of the handle); every call that reads results or rewrites inputs (upload, sync, fetch, ...) joins them first, so to the
 *            caller a run is one asynchronous operation, and back-to-back runs without such a call in between keep the two parts staggered
 *   sync   : wait for everything the handle has enqueued; returns CORB_ERR_OVERFLOW if any image overflowed
 *   fetch  : copy results of one image to host (synchronous) */
int corb_orb_upload(CorbOrb* h, int image, const uint8_t* img, int stride);
int corb_orb_run(CorbOrb* h, int n_images);
int corb_orb_sync(CorbOrb* h);
int corb_orb_fetch(CorbOrb* h, int image, CorbKeyPoint* keypoints, uint8_t* descriptors, int cap, int* n);
/* pre-quadtree candidate list of one level (cell-row-major order, ORBextractor.cc:789-829), for tests */
int corb_orb_fetch_candidates(CorbOrb* h, int image, int level, CorbKeyPoint* out, int cap, int* n);
/* host-buffer batches: n tightly packed images in one copy / all results of n images in one set of copies (pinned host memory = DMA) */
int corb_orb_upload_batch(CorbOrb* h, int first_image, int n_images, const uint8_t* imgs /* n x height x width */);
int corb_orb_capacity(CorbOrb* h);                  /* entries per image of the result arrays */
int corb_orb_fetch_batch(CorbOrb* h, int first_image, int n_images, CorbKeyPoint* keypoints /* [n][capacity] */, uint8_t* descriptors /* [n][capacity][32] */,
                         int32_t* counts /* [n] */);
/* device pointer + pitch of level-0 plane of slot `image` (to fill inputs without a host copy) */
int corb_orb_device_image(CorbOrb* h, int image, void** dptr, size_t* pitch);

/* ============================ stereo front-end =============================================
 * One client's per-frame work in Frame::Frame(stereo) (C/src/Frame.cc:61-117): left + right
 * ORBextractor::operator() and Frame::ComputeStereoMatches (C/src/Frame.cc:470-644), for a batch
 * of `max_frames` stereo frames per launch.  Frame f uses image slots 2f (left) and 2f+1 (right). */
typedef struct CorbStereo CorbStereo;
typedef struct CorbStereoConfig {
    CorbOrbConfig orb;      /* orb.max_images is ignored (= 2*max_frames) */
    int32_t max_frames;
    float fx;               /* Camera.fx */
    float bf;               /* Camera.bf  (Frame::mbf) */
} CorbStereoConfig;

int corb_stereo_create(const CorbStereoConfig* cfg, CorbStereo** out);
void corb_stereo_destroy(CorbStereo* h);
CorbOrb* corb_stereo_orb(CorbStereo* h);           /* the underlying batched extractor (borrowed) */
int corb_stereo_upload(CorbStereo* h, int frame, const uint8_t* left, const uint8_t* right, int stride);
int corb_stereo_run(CorbStereo* h, int n_frames);  /* async: extraction of 2n images + stereo match */
int corb_stereo_sync(CorbStereo* h);
/* whole batches with one copy each way; results strided by corb_orb_capacity(corb_stereo_orb(h)) */
int corb_stereo_upload_batch(CorbStereo* h, int first_frame, int n_frames, const uint8_t* left_right /* per frame: left image, right image */);
int corb_stereo_fetch_matches_batch(CorbStereo* h, int first_frame, int n_frames, float* u_right /* [n][capacity] */, float* depth, int32_t* n_matched /* [n] */);
/* mvuRight / mvDepth of the LEFT keypoints of `frame` (-1 = no match), n = left keypoint count */
int corb_stereo_fetch_matches(CorbStereo* h, int frame, float* u_right, float* depth, int cap, int* n, int* n_matched);

/* ---- the reference's operating point: a client hands over ONE stereo frame at a time (Frame::Frame(stereo), C/src/Frame.cc:61-117, called per frame by
 * Tracking::GrabImageStereo, C/src/Tracking.cc:166-203) ----
 * corb_stereo_frames: n frames (n small: 1, 2, 8 ...) host buffers in, host buffers out, in ONE call: one host-to-device transfer of the 2n images, the
 * kernel chain (replayed as a captured hipGraph per n), one pack kernel, ONE device-to-host transfer, ONE synchronisation.
 *   images : n x {left, right} x height x width bytes, tightly packed
 *   result : n blocks of corb_stereo_frame_layout().frame_bytes bytes; block f (frame f):
 *              int32 n_left, n_right, n_matched, status (0 = ok, else CORB_ERR_OVERFLOW of an image of the frame)
 *              CorbKeyPoint[capacity] left keypoints   at off_kp_left      uint8[capacity][32] left descriptors   at off_desc_left
 *              CorbKeyPoint[capacity] right keypoints  at off_kp_right     uint8[capacity][32] right descriptors  at off_desc_right
 *              float[capacity] mvuRight at off_u_right, float[capacity] mvDepth at off_depth (of the left keypoints, -1 = no match)
 *            only the first n_left / n_right entries of a section are written.
 * Both buffers should be page-locked (corb_pinned_alloc) so that the transfers are DMA; pageable memory works and is slower.
 * timing (may be NULL): device milliseconds of the three stages of this call (costs four event records). */
typedef struct CorbStereoFrameLayout {
    int32_t capacity, frame_bytes;
    int32_t off_kp_left, off_kp_right, off_desc_left, off_desc_right, off_u_right, off_depth;
} CorbStereoFrameLayout;
typedef struct CorbStereoFrameTiming { float ms_upload, ms_kernels, ms_download; } CorbStereoFrameTiming;
int corb_stereo_frame_layout(CorbStereo* h, CorbStereoFrameLayout* out);
int corb_stereo_frames(CorbStereo* h, int n_frames, const uint8_t* images, void* result, CorbStereoFrameTiming* timing);

/* per-kernel device timing (HIP events on the handle's own stream).  enable, run, sync, then read. */
typedef struct CorbKernelTime {
    char name[48];
    double total_ms;
    int64_t launches;
} CorbKernelTime;
int corb_orb_profile(CorbOrb* h, int enable);   /* 0 off; 1 time every launch of the product sequence (two overlapping half-batches);
                                                    2 as 1 but unsplit on one stream: stand-alone kernel durations */
int corb_orb_profile_read(CorbOrb* h, CorbKernelTime* out, int cap, int* n);   /* resets the accumulators */

/* ============================ descriptor matching ==========================================
 * Replaces the arithmetic of ORB_SLAM2::ORBmatcher (C/include/ORBmatcher.h:41-107). Flat arrays in,
 * indices out; the C++ adapter maps indices back to MapPoint*. Host pointers; synchronous. */

/* ORBmatcher::DescriptorDistance (C/src/ORBmatcher.cc:1792-1808) for n pairs a[i] vs b[i] */
int corb_descriptor_distance(const uint8_t* a, const uint8_t* b, int n, int32_t* dist, int device);

/* DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened: ascending node ids */
typedef struct CorbFeatVec {
    int32_t n_nodes;
    const uint32_t* node_id;
    const int32_t* offset;      /* n_nodes + 1 */
    const uint32_t* idx;        /* feature indices, offset[n_nodes] entries */
} CorbFeatVec;

typedef struct CorbBowSide {
    const uint8_t* desc;        /* n x 32 */
    const float* angle;         /* mvKeysUn[i].angle (only read if check_orientation) */
    const uint8_t* valid;       /* 1 = feature has a non-bad MapPoint (may be NULL for the Frame side) */
    int32_t n;
    CorbFeatVec fv;
} CorbBowSide;

/* variant 0: SearchByBoW(KeyFrame*,Frame&,...) (C/src/ORBmatcher.cc:162-291) and
 *            SearchByBoWInServer (294-423): match[iF] = KF feature index or -1, `match` has b.n entries
 * variant 1: SearchByBoW(KeyFrame*,KeyFrame*,...) (657-790): match[i1] = idx2 or -1, a.n entries
 * Returns the reference's return value (number of matches) in *n_matches. */
int corb_search_by_bow(int variant, const CorbBowSide* a, const CorbBowSide* b, float nnratio,
                       int check_orientation, int32_t* match, int* n_matches, int device);

typedef struct CorbTriSide {
    const uint8_t* desc;        /* n x 32 */
    const CorbKeyPoint* kp;     /* mvKeysUn */
    const float* u_right;       /* mvuRight */
    const uint8_t* has_mappoint;
    int32_t n;
    CorbFeatVec fv;
} CorbTriSide;

/* ORBmatcher::SearchForTriangulation (C/src/ORBmatcher.cc:792-958).  F12 row-major float 3x3;
 * (ex,ey) the epipole of KF1's centre in KF2 (:803-808); scale2/sigma2_2 = pKF2->mvScaleFactors /
 * mvLevelSigma2.  pairs: (idx1, idx2) sorted by idx1, capacity a.n pairs. */
int corb_search_for_triangulation(const CorbTriSide* a, const CorbTriSide* b, const float* F12, float ex, float ey,
                                  const float* scale2, const float* sigma2_2, int nlevels, int only_stereo,
                                  int check_orientation, int32_t* pairs, int* n_matches, int device);

/* ============================ projection-guided matchers (Tracking thread) ================
 * ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)  (C/src/ORBmatcher.cc:45-131, TrackLocalMap) and
 * ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)       (C/src/ORBmatcher.cc:1470-1614, TrackWithMotionModel),
 * including Frame::AssignFeaturesToGrid / GetFeaturesInArea (C/src/Frame.cc:230-245, 331-384).  Both are greedy and order
 * dependent in the reference (a feature can be claimed once); the results here are identical to the sequential order. */
typedef struct CorbFrameView {       /* what the matchers read from the current Frame */
    const CorbKeyPoint* keys_un; const float* u_right; const uint8_t* desc; int32_t n;
    const uint8_t* claimed;          /* mvpMapPoints[i] holds a MapPoint with Observations()>0 */
    float min_x, min_y, max_x, max_y;            /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    const float* scale; int32_t nlevels;         /* mvScaleFactors */
} CorbFrameView;
typedef struct CorbTrackedPoint {    /* MapPoint fields set by Frame::isInFrustum */
    float proj_x, proj_y, proj_xr, view_cos;     /* mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos */
    int32_t level;                   /* mnTrackScaleLevel */
    uint8_t valid;                   /* mbTrackInView && !isBad() */
    uint8_t claims;                  /* Observations()>0 */
    uint8_t pad[2];
} CorbTrackedPoint;
typedef struct CorbLastPoint {       /* one feature of the last frame */
    float world[3];                  /* pMP->GetWorldPos() */
    float angle;                     /* LastFrame.mvKeysUn[i].angle */
    int32_t octave;                  /* LastFrame.mvKeys[i].octave */
    uint8_t valid;                   /* has a MapPoint && !mvbOutlier[i] */
    uint8_t claims;                  /* pMP->Observations()>0 */
    uint8_t pad[2];
} CorbLastPoint;
/* match[i] per current-frame feature = index of the assigned map point / last-frame feature or -1; *n_matches = return value */
int corb_search_by_projection_map(const CorbFrameView* frame, const CorbTrackedPoint* points, const uint8_t* point_desc /* n x 32, pMP->GetDescriptor() */,
                                  int n_points, float th, float nnratio, int32_t* match, int* n_matches, int device);
int corb_search_by_projection_frame(const CorbFrameView* cur, const float* Tcw /* 16, current pose */, const float* Tlw /* 16, last pose */,
                                    float fx, float fy, float cx, float cy, float bf, float mb, const CorbLastPoint* last,
                                    const uint8_t* last_desc /* n x 32, pMP->GetDescriptor() */, int n_last, float th, int mono,
                                    int check_orientation, int32_t* match, int* n_matches, int device);

/* ---- matchers that project MapPoints into a KeyFrame, or into the current Frame for relocalisation ----
 * The adapter evaluates the pointer-level skip tests (NULL / isBad() / sAlreadyFound / IsInKeyFrame / vbAlreadyMatched) into
 * `valid`, and applies the pointer-level consequences of a match (Replace / AddObservation / AddMapPoint) from the returned
 * indices; everything arithmetic (projection, distance and viewing-angle gates, PredictScale, grid search, octave window,
 * chi2 gate, Hamming minimum, rotation histogram) runs on the device. */
typedef struct CorbKeyFrameView {    /* KeyFrame (Fuse, SearchBySim3) or the current Frame (relocalisation) */
    const CorbKeyPoint* keys_un; const float* u_right; const uint8_t* desc; int32_t n;   /* mvKeysUn, mvuRight, mDescriptors */
    float min_x, min_y, max_x, max_y;            /* mnMinX .. mnMaxY */
    const float* scale; const float* inv_level_sigma2; int32_t nlevels;   /* mvScaleFactors, mvInvLevelSigma2 */
    float log_scale_factor;                      /* mfLogScaleFactor */
    float fx, fy, cx, cy, bf;
} CorbKeyFrameView;
typedef struct CorbMapPointView {
    float world[3];                  /* GetWorldPos() */
    float normal[3];                 /* GetNormal() (Fuse) */
    float min_distance, max_distance;/* mfMinDistance, mfMaxDistance (Get*DistanceInvariance() = 0.8f / 1.2f times these) */
    float angle;                     /* relocalisation: pKF->mvKeysUn[i].angle of the keyframe feature holding the point */
    uint8_t valid; uint8_t pad[3];
} CorbMapPointView;
/* int SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize) (C/src/ORBmatcher.cc:540-655): the monocular
 * initialiser's matcher (Tracking::MonocularInitialization, C/src/Tracking.cc:606).  f1 / f2: keys_un, desc, n and (f2) the image bounds are read (u_right, claimed, scale
 * are not).  prev_matched: n(f1) x 2 floats, vbPrevMatched, read and written; matches12[i1] = feature of F2 or -1; nnratio = mfNNratio, check_orientation =
 * mbCheckOrientation; *n_matches = the return value. */
int corb_search_for_initialization(const CorbFrameView* f1, const CorbFrameView* f2, float* prev_matched, int window_size, float nnratio, int check_orientation,
                                   int32_t* matches12, int* n_matches, int device);
/* int SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>& sAlreadyFound, th, ORBdist) (C/src/ORBmatcher.cc:1616-1744).
 * claimed[i] = CurrentFrame.mvpMapPoints[i] holds a MapPoint; match[i] per current-frame feature = point index or -1. */
int corb_search_by_projection_reloc(const CorbKeyFrameView* cur, const uint8_t* claimed, const float* Tcw /* 16 */, const CorbMapPointView* points,
                                    const uint8_t* point_desc /* n x 32 */, int n_points, float th, int orb_dist, int check_orientation,
                                    int32_t* match, int* n_matches, int device);
/* int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th) (C/src/ORBmatcher.cc:425-538;
 * callers: C/src/LoopClosing.cc:377, S/src/GlobalOptimize.cpp:199 -- the last matcher before CorrectLoop / the global BA of a fusion event).
 * claimed[idx] = vpMatched[idx] != NULL on entry (NULL = no feature holds a point); points[i].valid = !pMP->isBad() && !spAlreadyFound.count(pMP);
 * match[idx] per keyframe feature = index of the point this call writes into vpMatched[idx], or -1; *n_matches = the return value. */
int corb_search_by_projection_scw(const CorbKeyFrameView* kf, const uint8_t* claimed, const float* Scw /* 16 */, const CorbMapPointView* points,
                                  const uint8_t* point_desc /* n x 32 */, int n_points, float th, int32_t* match, int* n_matches, int device);
/* int Fuse(KeyFrame*, const vector<MapPoint*>&, th) (:960-1116): sim3 = 0, T = Tcw (16), Ow = pKF->GetCameraCenter();
 * int Fuse(KeyFrame*, cv::Mat Scw, vpPoints, th, vpReplacePoint) (:1118-1241): sim3 = 1, T = Scw (16), Ow ignored.
 * best_idx[i] = keyframe feature into which point i is fused (bestDist <= TH_LOW) or -1; *n_fused = the return value. */
int corb_fuse(const CorbKeyFrameView* kf, const float* T, const float* Ow, int sim3, const CorbMapPointView* points, const uint8_t* point_desc,
              int n_points, float th, int32_t* best_idx, int32_t* best_dist, int* n_fused, int device);
/* int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, s12, R12, t12, th) (:1244-1468).
 * points1[i] / points2[i] = the MapPoint of feature i of KF1 / KF2 (valid = exists, !isBad(), not already matched);
 * match12[i1] = feature of KF2 whose MapPoint becomes vpMatches12[i1], or -1; *n_found = the return value. */
int corb_search_by_sim3(const CorbKeyFrameView* kf1, const CorbKeyFrameView* kf2, const float* T1w, const float* T2w,
                        const CorbMapPointView* points1, const uint8_t* desc1, const CorbMapPointView* points2, const uint8_t* desc2,
                        float s12, const float* R12 /* 9 */, const float* t12 /* 3 */, float th, int32_t* match12, int* n_found, int device);

/* ---- map maintenance next to the hot path ----
 * void MapPoint::ComputeDistinctiveDescriptors() (C/src/MapPoint.cc:337-402) for a batch of map points: desc = the descriptors
 * of every point's non-bad observations stacked in std::map order, point p owns rows offset[p] .. offset[p+1];
 * best_idx[p] = row (relative to offset[p]) the point adopts as mDescriptor, -1 for a point without descriptors. */
int corb_distinctive_descriptors(const uint8_t* desc, const int32_t* offset, int n_points, int32_t* best_idx, int device);
/* void MapFusion::insertServerMapToGlobleMap(ServerMap*, cv::Mat To2n) (S/src/MapFusion.cpp:622-658), arithmetic part: every
 * keyframe pose Tcw <- Tcw * To2n (4x4 row-major, in place), every map point p <- Rwc (p - tcw) with To2n = [Rcw | tcw]. */
int corb_rebase_map(const float* To2n, float* poses, int n_poses, float* points, int n_points, int device);

/* ============================ global bundle adjustment =====================================
 * Replaces the arithmetic of Optimizer::GlobalBundleAdjustemnt -> BundleAdjustment
 * (C/src/Optimizer.cc:43-270) and the g2o pieces it drives: EdgeSE3ProjectXYZ /
 * EdgeStereoSE3ProjectXYZ (G/types/types_six_dof_expmap.{h,cpp}), BlockSolver_6_3 Schur solve
 * (G/core/block_solver.hpp:354-604), Levenberg-Marquardt (G/core/optimization_algorithm_levenberg.cpp).
 * The adapter flattens KeyFrames / MapPoints and applies the nLoopKF write-back policy. */
typedef struct CorbBAEdge {
    int32_t pose, point;
    float u, v, u_right;        /* u_right < 0 : monocular 2-D edge, else stereo 3-D edge */
    float inv_sigma2;           /* pKF->mvInvLevelSigma2[kpUn.octave] */
} CorbBAEdge;

typedef struct CorbBAProblem {
    int32_t n_poses, n_points, n_edges;
    const float* poses;         /* n_poses x 16 row-major Tcw (cv::Mat CV_32F) */
    const uint8_t* pose_fixed;  /* mnId==1 || getFixed() */
    const float* points;        /* n_points x 3 */
    const uint8_t* point_fixed;
    const CorbBAEdge* edges;
    float fx, fy, cx, cy, bf;   /* shared camera, used when intr == NULL (single-client maps) */
    const float* intr;          /* n_poses x 5: fx, fy, cx, cy, bf of EVERY keyframe -- the reference sets them per edge from the observing keyframe
                                   (e->fx = pKF->fx; ... e->bf = pKF->mbf, C/src/Optimizer.cc:160-163, 189-193; KeyFrame.h:68 serialises them per keyframe), so a
                                   fused map of clients with different cameras (KITTI00-02.yaml vs KITTI04-12.yaml) is one problem.  NULL = shared camera above. */
} CorbBAProblem;

typedef struct CorbBAResult {
    float* poses;               /* n_poses x 16 (fixed poses copied through) */
    float* points;              /* n_points x 3 */
    double* chi2;               /* iterations+1 entries: initial, then after each outer iteration (may be NULL) */
    double* lambda;             /* iterations entries (may be NULL) */
    int32_t iters_done;
    int32_t trials_total;
    double ms_total, ms_build, ms_schur, ms_solve, ms_update;   /* device time of the call; the phase times are measured from 65 536 observations on (0 below) */
    int32_t solver_used;        /* 1 dense Cholesky (csrc/dense_chol.hip; inside the one-workgroup optimiser for small problems), 2 block-sparse PCG,
                                   3 fused single-pose kernel (6x6 LDL^T on the device) */
    int32_t pcg_iterations;     /* total CG iterations over all LM trials */
    /* structure of the last optimize() call (sizes behind the roofline figures of bench.py) */
    int32_t free_poses, free_points, active_edges;
    int64_t nnz_blocks;         /* 6x6 blocks of the reduced camera system, both triangles (0 for the fused small-problem kernel) */
    int64_t schur_pairs;        /* (edge, edge) pairs of the Schur complement = sum over the upper blocks of their co-observed landmarks */
    int32_t pc_block;           /* poses per block of the block-Jacobi preconditioner actually used (PCG) */
    int32_t pc_levels;          /* coarse levels of the multilevel preconditioner (0 = block Jacobi only) */
    /* self-certification of a call that took the PCG solver (CORB_ABI_VERSION >= 5).  g2o's reduced solve is an exact factorisation
       (G/solvers/linear_solver_eigen.h:94-124: residual at rounding level); the iterative solve reports what it reached instead: */
    double pcg_residual_max;    /* largest TRUE relative residual |b - S x| / |b| over the call's reduced solves, recomputed in FP64 by an independent kernel
                                   after each solve (not the recurrence's residual); 0 when no PCG solve ran */
    double pcg_residual_last;   /* the same of the last solve */
    double grad_inf;            /* |J' Omega r|_inf (poses and map points) of a linearisation at the returned estimates; < 0: not computed (dense / small paths) */
    int32_t pcg_refined_trials; /* default tolerance policy (CorbBAOptions.pcg_tol == 0): LM trials whose solve was continued from 1e-6 to 1e-8 before the trial was decided */
    int32_t reserved0;          /* staged calls (corb_ba_solve_staged, corb_local_ba_store): 1 = the window was flattened, optimised and classified on the device,
                                   0 = the host flattening ran -- same results; informational */
} CorbBAResult;

/* linear solver for the reduced camera system (replaces g2o::LinearSolverEigen, G/solvers/linear_solver_eigen.h:94-124) */
typedef struct CorbBAOptions {
    int32_t solver;             /* 0 auto (dense up to 256 free poses -- up to 16 free poses and 2 048 observations the whole optimisation runs in one
                                   workgroup with its own in-LDS Cholesky, above that the blocked Cholesky of dense_chol.hip --, PCG above 256; staged problems with ONE free pose and fixed points:
                                   the fused single-workgroup kernel), 1 dense Cholesky, 2 block-sparse PCG, 3 fused single-pose kernel */
    double  pcg_tol;            /* relative residual |r|/|b| at which CG stops.  > 0: that tolerance for every solve of the call.  0 (default): 1e-8 on problems of up to 256
                                   free keyframes; above, a forcing sequence -- 1e-6 in the first iteration, then clamp(1e-2 x the previous iteration's relative chi2 gain,
                                   1e-8, 1e-6) -- and a trial whose accept / reject decision or lambda factor could depend on the accuracy of its solve (rho near 0, in the
                                   steep part 0.80 .. 0.97 of the lambda schedule, or a predicted decrease below 3e-6 chi2) has the SAME solve continued to 1e-8 before it is
                                   decided.  chi2 after every iteration within 5e-8 relative of a 1e-13 solve on 320 ... 20 000 keyframes (the parity bar is 1e-4; what sets
                                   1e-6 is lambda: see csrc/corb_ba.cpp BAChoice) -- profiles/r05_pcg_tol_sweep.txt */
    int32_t pcg_max_iter;       /* default 4000; not converged => the LM trial is rejected like a failed factorisation */
    int32_t pc_block;           /* poses per block of the block-Jacobi preconditioner: 0 = auto (1 below 128 free poses, 16 above), 1 = the 6x6 diagonal blocks, 8 or 16
                                   (dense diagonal blocks inverted on every 3rd accepted LM trial and after a rejected one) */
    int32_t pc_multilevel;      /* coarse levels next to the 16-pose blocks (linear hats over the keyframe order, stride 8 then 4, Galerkin matrices, block Jacobi per level:
                                   csrc/ba_multilevel.h): 0 = auto (on from 256 free poses: every map the PCG solver takes by default), 1 = off, 2 = on (needs pc_block 16 or auto with >= 128 free poses) */
    float   scale_factor;       /* corb_ba_solve_store with loop_kf == 0 only: ORBextractor's scaleFactor (1.2 in every reference yaml).  > 0: SetWorldPos is followed by
                                   MapPoint::UpdateNormalAndDepth on the records (see corb_ba_solve_store); 0 (default): normal / min_distance / max_distance are left alone.
                                   (The field fills what was padding: sizeof(CorbBAOptions) is unchanged.) */
} CorbBAOptions;

/* optimizer.optimize(nIterations) with bRobust / pbStopFlag semantics of Optimizer.cc:54-270 */
int corb_ba_solve(const CorbBAProblem* problem, int iterations, int robust, volatile int* stop_flag,
                  CorbBAResult* result, int device);
int corb_ba_solve_ex(const CorbBAProblem* problem, int iterations, int robust, volatile int* stop_flag,
                     CorbBAResult* result, int device, const CorbBAOptions* options /* NULL = defaults */);

/* corb_ba_solve_ex with the graph flattening done ON THE DEVICE (the path corb_ba_solve_store takes: ba_flatten.hip) instead of on the host: the host arrays are
 * uploaded as they are, grouped by map point, and indexed / sorted / patterned by kernels.  Same lists as the host flattening, element for element (the tests
 * compare the two); a caller with host arrays has no reason to prefer it except at the largest sizes, where it saves the host-side list building. */
int corb_ba_solve_devflat(const CorbBAProblem* problem, int iterations, int robust, CorbBAResult* result, int device, const CorbBAOptions* options);

/* The dense solver of the reduced camera system on its own (what g2o::LinearSolverEigen / LinearSolverDense do for a dense block matrix,
 * G/solvers/linear_solver_eigen.h:94-124): A x = b for a symmetric positive definite n x n matrix (row-major, only the lower triangle is read), hand-written
 * blocked Cholesky on the FP64 matrix cores (csrc/dense_chol.hip).  Host pointers; *info = 0 or 1 + the first column whose pivot is not positive (x is then
 * meaningless).  Exposed for tests and for adapters with their own small dense systems. */
int corb_spd_solve(const double* A, int n, const double* b, double* x, int* info, int device);

/* One optimize() call plus the outlier test that follows it.  Sequences of stages express
 *   Optimizer::LocalBundleAdjustment (C/src/Optimizer.cc:487-838): {5, robust, 5.991, 7.815, check_depth=1},
 *                                                                   {10, non-robust, 5.991, 7.815, check_depth=1, allow_reactivate=1}
 *                                     (the final "Check inlier observations" pass tests EVERY edge, also those switched off after the first round, with its
 *                                      last computed chi2 and a fresh depth, :763-790).  pbStopFlag: raised before the first optimize() -> outputs = inputs, no
 *                                      outliers (:706-708); raised later -> the remaining optimize() calls are skipped and the LAST stage's test is applied
 *                                      (bDoMore = false skips only the second round, :712-716).
 *   Optimizer::PoseOptimization     (C/src/Optimizer.cc:272-485): 4 x {10, robust (last: non-robust), 5.991, 7.815,
 *                                     recompute_inactive=1, allow_reactivate=1, reset_estimates=1, float_compare=1}
 * chi2 of an ACTIVE edge is the value of its last computeError() inside optimize() (g2o does not refresh it afterwards;
 * after a rejected last trial it belongs to the rejected state) -- reproduced here; depth is evaluated fresh. */
typedef struct CorbBAStage {
    int32_t iterations, robust;
    float chi2_mono, chi2_stereo;
    int32_t check_depth;            /* edge also becomes an outlier if depth <= 0 (isDepthPositive) */
    int32_t recompute_inactive;     /* computeError() on inactive edges before the test */
    int32_t allow_reactivate;       /* inactive edges may become active (inlier) again */
    int32_t reset_estimates;        /* restart from the input estimates */
    int32_t float_compare;          /* `const float chi2 = e->chi2()` comparison */
    float huber_mono, huber_stereo; /* Huber deltas, (float)sqrt(5.991) / (float)sqrt(7.815) in both callers */
} CorbBAStage;
/* edge_outlier[n_edges]: 1 = classified outlier after the last stage (vToErase / mvbOutlier).  Vertices that never had an
 * active edge are passed through unchanged. */
int corb_ba_solve_staged(const CorbBAProblem* problem, const CorbBAStage* stages, int n_stages, volatile int* stop_flag,
                         CorbBAResult* result, uint8_t* edge_outlier, int device, const CorbBAOptions* options);

/* int Optimizer::PoseOptimization(Frame* pFrame) (C/src/Optimizer.cc:272-485) for a BATCH of frames (the tracking threads of
 * many clients): per frame the 4 x 10 Levenberg-Marquardt rounds with re-classification of the observations between the
 * rounds run inside ONE workgroup without host round trips.  Per observation i of a frame: pMP->GetWorldPos(),
 * mvKeysUn[i].pt, mvuRight[i] (< 0 = monocular edge), mvInvLevelSigma2[octave]. */
typedef struct CorbPoseOptFrame {
    const float* Tcw;           /* 4x4 row-major start pose (pFrame->mTcw) */
    int32_t n_obs;
    const float* points;        /* n_obs x 3 */
    const float* u; const float* v; const float* u_right;
    const float* inv_sigma2;
    float fx, fy, cx, cy, bf;
} CorbPoseOptFrame;
/* Tcw_out: n_frames x 16 (pFrame->SetPose); outlier[f] (may be NULL): n_obs flags = pFrame->mvbOutlier; n_inliers[f] =
 * nInitialCorrespondences - nBad (the function's return value) */
int corb_pose_optimization_batch(const CorbPoseOptFrame* frames, int n_frames, float* Tcw_out, uint8_t* const* outlier,
                                 int32_t* n_inliers, int device);

/* int Optimizer::OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12, float th2, bool bFixScale)
 * (C/src/Optimizer.cc:1119-1311) for a batch of loop-closure candidates, one workgroup each: optimize(5), chi2 classification,
 * optimize(5 or 10), final classification -- with g2o's numeric Jacobians (delta 1e-9) of EdgeSim3ProjectXYZ /
 * EdgeInverseSim3ProjectXYZ, Huber(sqrt(th2)), LinearSolverDense.  Per correspondence i of a candidate: P3D1c = R1w*P3D1w+t1w and
 * P3D2c (camera-frame points as the adapter already computes them), the two undistorted keypoints, mvInvLevelSigma2 of their octaves. */
typedef struct CorbSim3Problem {
    int32_t n;
    const float* p1c; const float* p2c;              /* n x 3 */
    const float* obs1; const float* obs2;            /* n x 2 */
    const float* inv_sigma2_1; const float* inv_sigma2_2;
    float fx1, fy1, cx1, cy1, fx2, fy2, cx2, cy2;    /* pKF1->mK, pKF2->mK */
} CorbSim3Problem;
/* R12 (n_problems x 9, row-major), t12 (x 3), s12: g2oS12 in / out (unchanged for a candidate that keeps fewer than 10 correspondences
 * after the first round; its n_inliers is 0).  removed[f][i] = 1: vpMatches1[idx] is nulled.  n_inliers[f] = the return value. */
int corb_optimize_sim3(const CorbSim3Problem* problems, int n_problems, double* R12, double* t12, double* s12, float th2, int fix_scale,
                       uint8_t* const* removed, int32_t* n_inliers, int32_t* iterations /* may be NULL */, int device);

/* void Optimizer::OptimizeEssentialGraph(Cache*, KeyFrame* pLoopKF, KeyFrame* pCurKF, NonCorrectedSim3, CorrectedSim3, LoopConnections, bFixScale)
 * (C/src/Optimizer.cc:840-1117).  The adapter builds the graph exactly as lines 866-1037 do: one g2o::Sim3 per non-bad keyframe
 * (S = quaternion x y z w, translation, scale; CorrectedSim3 value or Sim3(Rcw, tcw, 1)), fixed = (pKF == pLoopKF || getFixed()), and per
 * EdgeSim3 the pair (vertex 0 = vi, vertex 1 = vj) with its measurement Sji = Sjw * Swi.  The library runs optimize(iterations = 20) with
 * lambda_init 1e-16 (numeric Jacobians, identity information), then the SE3 recovery Tiw = [R | t/s] (Tiw_out, K x 16, may be NULL) and the
 * map point correction p <- correctedSwr.map(Srw.map(p)) for points whose reference keyframe index point_ref[m] >= 0 (in place).
 * S is updated in place; chi2_hist (iterations + 1, may be NULL); *iters_done optional. */
int corb_optimize_essential_graph(int n_keyframes, double* S, const uint8_t* fixed, int n_edges, const int32_t* vi, const int32_t* vj,
                                  const double* measurement, int iterations, int fix_scale, float* Tiw_out, int n_points,
                                  const int32_t* point_ref, float* points, double* chi2_hist, int32_t* iters_done, int device);

/* ============================ device-resident keyframe store + map push =====================
 * What the reference serialises per KeyFrame for the client -> server push (C/include/KeyFrame.h:59-87: mnId, mvKeys / mvKeysUn, mDescriptors, mvuRight,
 * mvDepth, mBowVec / mFeatVec ...; boost text archives through ROS services every 6 s, C/src/Cache.cc:322-375, C/src/DataDriver.cc:135-193) kept as one
 * fixed-size SoA record per keyframe IN DEVICE MEMORY: 28-byte keypoints, 32-byte descriptors, mvuRight, mvDepth, the keypoint angles, the "has a good
 * MapPoint" flags and the DBoW2 FeatureVector groups.  A slot is filled device-to-device from the stereo front-end's results (no host trip), matched
 * against other slots without uploads, and pushed to the server rank with RCCL send / receive on the device buffers (one process per GPU over xGMI). */
typedef struct CorbKfStore CorbKfStore;
int corb_kf_store_create(int device, int capacity_keyframes, int max_features, CorbKfStore** out);
void corb_kf_store_destroy(CorbKfStore* s);
int corb_kf_store_record_bytes(const CorbKfStore* s);                 /* bytes of one slot record (what a push moves per keyframe) */
/* slot <- left keypoints / descriptors / mvuRight / mvDepth of frame `frame` of a stereo front-end after corb_stereo_run (device-to-device, asynchronous on
 * the front-end's stream; the store's later calls wait for it) */
int corb_kf_store_put_from_stereo(CorbKfStore* s, int slot, CorbStereo* sf, int frame, uint64_t keyframe_id);
/* slot <- host arrays (adapters / tests); any pointer except kp / desc may be NULL (u_right, depth default to -1) */
int corb_kf_store_put_host(CorbKfStore* s, int slot, const CorbKeyPoint* kp, const uint8_t* desc, const float* u_right, const float* depth, int n, uint64_t keyframe_id);
/* the parts the host computes: DBoW2 FeatureVector (Frame::ComputeBoW, C/src/Frame.cc:397-406) and the per-feature "vpMapPoints[i] && !isBad()" flags */
int corb_kf_store_set_bow(CorbKfStore* s, int slot, const CorbFeatVec* fv);
int corb_kf_store_set_flags(CorbKfStore* s, int slot, const uint8_t* has_good_mappoint /* n entries, NULL = all 0 */);
/* slot -> host (any output may be NULL; *n = feature count); fv arrays need max_features (+1 for the offsets) entries */
int corb_kf_store_get(CorbKfStore* s, int slot, CorbKeyPoint* kp, uint8_t* desc, float* u_right, float* depth, uint8_t* flags, int cap, int* n, uint64_t* keyframe_id,
                      uint32_t* fv_node_id, int32_t* fv_offset, uint32_t* fv_idx, int32_t* fv_n_nodes);
/* corb_search_by_bow on two slots (possibly of two stores on the same device): nothing is uploaded but the short list of common vocabulary nodes.
 * variant 0: match has n(slot_b) entries, variant 1: n(slot_a) entries (see corb_search_by_bow). */
int corb_search_by_bow_slots(int variant, CorbKfStore* a, int slot_a, CorbKfStore* b, int slot_b, float nnratio, int check_orientation,
                             int32_t* match, int* n_matches);
/* corb_search_for_triangulation on two slots (has_mappoint = the slots' flags) */
int corb_search_for_triangulation_slots(CorbKfStore* a, int slot_a, CorbKfStore* b, int slot_b, const float* F12, float ex, float ey,
                                        const float* scale2, const float* sigma2_2, int nlevels, int only_stereo, int check_orientation,
                                        int32_t* pairs, int* n_matches);

/* ---- the rest of the push payload: poses, intrinsics, per-feature map points (KeyFrame.h:65-79) and the MapPoint records (MapPoint.h:52-72) ---- */
#define CORB_KF_BAD    1u        /* mbBad */
#define CORB_KF_FIXED  2u        /* ifFixed (Cache.cc:482: entities received from the server) */
#define CORB_MP_BAD    1u
#define CORB_MP_FIXED  2u
#define CORB_NO_MAP_POINT 0xFFFFFFFFFFFFFFFFull
typedef struct CorbKeyFrameMeta {       /* header of a keyframe record */
    uint64_t id;                        /* mnId (globally unique: client c counts from (c-1)*1000000+1, KeyFrame.cc:49) */
    int32_t client_id;                  /* mnClientId */
    uint32_t flags;                     /* CORB_KF_BAD | CORB_KF_FIXED */
    float fx, fy, cx, cy, bf;           /* fx, fy, cx, cy, mbf */
    int32_t nlevels;                    /* mnScaleLevels */
    float Tcw[16];                      /* Tcw, row-major 4x4 */
    float TcwGBA[16];                   /* mTcwGBA */
    uint64_t ba_global_for_kf;          /* mnBAGlobalForKF */
    float inv_level_sigma2[16];         /* mvInvLevelSigma2 */
} CorbKeyFrameMeta;
int corb_kf_store_set_meta(CorbKfStore* s, int slot, const CorbKeyFrameMeta* meta);      /* also sets the record's id */
/* corb_kf_store_put_host + corb_kf_store_set_meta as ONE upload and one synchronisation: a tracked Frame enters its record (per-frame use: corb_track_*) */
int corb_kf_store_put_frame(CorbKfStore* s, int slot, const CorbKeyPoint* kp, const uint8_t* desc, const float* u_right, const float* depth, int n, const CorbKeyFrameMeta* meta);
int corb_kf_store_get_meta(CorbKfStore* s, int slot, CorbKeyFrameMeta* meta);
/* mvpMapPoints as ids (LightMapPoint::mnMapPointId, KeyFrame.h:78): n(slot) entries, CORB_NO_MAP_POINT = none */
int corb_kf_store_set_map_points(CorbKfStore* s, int slot, const uint64_t* mp_id);
int corb_kf_store_get_map_points(CorbKfStore* s, int slot, uint64_t* mp_id, int cap);

/* slots first .. first+n <- n keyframes from host arrays in ONE upload and ONE kernel (adapters that flatten a whole map, benches): meta[n]; the features of
 * keyframe i are entries feat_offset[i] .. feat_offset[i+1] of kp / desc / u_right / depth / mp_id (desc, u_right, depth, mp_id may be NULL: zero descriptors,
 * -1, -1, CORB_NO_MAP_POINT).  Flags and BoW groups of the slots are cleared like corb_kf_store_put_host does. */
int corb_kf_store_put_batch(CorbKfStore* s, int first, int n, const CorbKeyFrameMeta* meta, const int32_t* feat_offset, const CorbKeyPoint* kp, const uint8_t* desc,
                            const float* u_right, const float* depth, const uint64_t* mp_id);

typedef struct CorbMapPointRecord {     /* header of a map-point record; the observations follow it in the record */
    uint64_t id;                        /* mnId */
    uint64_t ref_kf_id;                 /* mpRefKF->mnId */
    uint8_t descriptor[32];             /* mDescriptor; byte offset 16: the kernels read it as four aligned 64-bit words */
    int32_t client_id;                  /* mnClientId */
    int32_t n_obs;                      /* mObservations.size() */
    uint32_t flags;                     /* CORB_MP_BAD | CORB_MP_FIXED */
    float world_pos[3];                 /* mWorldPos */
    float normal[3];                    /* mNormalVector */
    float min_distance, max_distance;   /* mfMinDistance, mfMaxDistance */
    float pos_gba[3];                   /* mPosGBA */
    uint64_t ba_global_for_kf;          /* mnBAGlobalForKF */
} CorbMapPointRecord;                   /* 112 bytes; this is also the on-wire record of corb_map_push_ex */
typedef struct CorbMpStore CorbMpStore;
/* one fixed-size record per MapPoint in device memory: the header above + up to max_observations (keyframe id, feature index) pairs = mObservations */
int corb_mp_store_create(int device, int capacity_points, int max_observations, CorbMpStore** out);
void corb_mp_store_destroy(CorbMpStore* s);
int corb_mp_store_record_bytes(const CorbMpStore* s);
/* slots first .. first+n <- host records; the observations of record i are obs_kf_id / obs_feature_idx [obs_offset[i] .. obs_offset[i+1]) in mObservations order
 * (ascending keyframe id).  A record with more than max_observations observations => CORB_ERR_CAPACITY, nothing written. */
int corb_mp_store_put_host(CorbMpStore* s, int first, int n, const CorbMapPointRecord* records, const int32_t* obs_offset, const uint64_t* obs_kf_id, const uint32_t* obs_feature_idx);
/* slots -> host: records[n], and (optional) the observations padded to max_observations per record: obs_kf_id / obs_feature_idx [n][max_observations] */
int corb_mp_store_get(CorbMpStore* s, int first, int n, CorbMapPointRecord* records, uint64_t* obs_kf_id, uint32_t* obs_feature_idx);

/* RCCL communicator of the client / server ranks (one process per GPU).  Rank 0 of the job calls corb_comm_unique_id and hands the 128 bytes to every rank
 * by its own means (torch.distributed broadcast, a ROS parameter, a file); every rank then calls corb_comm_create.  librccl.so is loaded on first use
 * (its version is checked: the ncclDataType_t constants used here are those of RCCL 2.x). */
typedef struct CorbComm CorbComm;
int corb_comm_unique_id(void* id128);
int corb_comm_create(const void* id128, int rank, int world, int device, CorbComm** out);
/* The same communicator interface over an in-process transport: `world` handles that share a mailbox, each to be driven by its own host thread; records
 * travel as device-to-device copies.  For servers that run several clients' stores in one process, and for exercising the N-rank push on one GPU.
 * devices: [world] HIP device of every rank, NULL = all on device 0. */
int corb_comm_create_local(int world, const int* devices, CorbComm** out /* [world] */);
void corb_comm_destroy(CorbComm* c);
int corb_comm_rank(const CorbComm* c);
int corb_comm_world(const CorbComm* c);

/* Map push (replaces the insertKeyFrameToMap / insertMapPointToMap service batches, C/src/DataDriver.cc:135-193, S/src/MapFusion.cpp:31-190): every rank
 * calls it collectively; rank r sends the keyframe records kf_slots[0..n_kf) and the map-point records mp_slots[0..n_mp) of its stores to `root`, which
 * files the records of rank r in its own stores from kf_dst_first[r] / mp_dst_first[r] on (rank order, its own included).
 * The call is collective-safe: every rank first contributes a header (its local argument verdict, counts, record sizes) to an all-gather; the root checks
 * capacity and placement with corb_map_push_plan and its verdict travels in a second all-gather, so on ANY rank's error EVERY rank returns the same
 * error code before a single record is sent -- no rank is left waiting in a send.  Records travel as one message per rank and store (a rank's selected
 * records are packed into a contiguous staging buffer first, so source and destination slots of the root may overlap). */
typedef struct CorbMapPush {
    CorbKfStore* kf; const int32_t* kf_slots; int32_t n_kf;
    CorbMpStore* mp; const int32_t* mp_slots; int32_t n_mp;          /* mp may be NULL on every rank: keyframes only */
    const int32_t* kf_dst_first; const int32_t* mp_dst_first;        /* root: [world] */
    int32_t* kf_recv_counts; int32_t* mp_recv_counts;                /* root, optional: [world] */
} CorbMapPush;
int corb_map_push_ex(CorbComm* c, const CorbMapPush* push, int root);
/* keyframes only (round 2's signature) */
int corb_map_push(CorbComm* c, CorbKfStore* s, const int* slots, int n_slots, int root, const int* dst_first, int* recv_counts);
/* The push's bookkeeping as a pure function (no device, no communicator; callable on a CPU-only box): what every rank contributed to the header
 * all-gather -> the verdict every rank returns.  CORB_OK, or: a rank's local status != 0 => that status; record sizes differ from the root's =>
 * CORB_ERR_ARG; a destination range outside the root's capacity => CORB_ERR_CAPACITY; two ranks' destination ranges overlap => CORB_ERR_ARG.
 * *failing_rank = the first rank the verdict is about (-1 if none). */
typedef struct CorbPushHeader { int32_t status, n_kf, n_mp, kf_record_bytes, mp_record_bytes; } CorbPushHeader;
int corb_map_push_plan(int world, int root, const CorbPushHeader* headers /* [world] */, int kf_capacity, int mp_capacity,
                       const int32_t* kf_dst_first, const int32_t* mp_dst_first, int* failing_rank);

/* The messages of a push as pure arithmetic on the gathered headers (CPU-testable; what corb_map_push_ex / corb_map_push_begin post): rank `rank` SENDS one message
 * per store with records to `root` (kind 0 = keyframes, 1 = map points; first_record = 0: the packed staging buffer), the root RECEIVES one per rank and store into
 * its slots from kf_dst_first[r] / mp_dst_first[r] on (first_record = that slot), in rank order, keyframes before map points -- the order in which the sender
 * posted them: messages between a pair of ranks match in posting order.  sends / recvs: room for 2 / 2 * world entries.  Returns CORB_OK or CORB_ERR_ARG. */
typedef struct CorbPushMsg { int32_t peer, kind, first_record, n_records; int64_t bytes; } CorbPushMsg;
int corb_map_push_messages(int world, int rank, int root, const CorbPushHeader* headers /* [world] */, const int32_t* kf_dst_first, const int32_t* mp_dst_first,
                           CorbPushMsg* sends, int* n_sends, CorbPushMsg* recvs, int* n_recvs);
/* the four RCCL entry points the record exchange calls (signatures of ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd with comm / stream as void*).
 * corb_comm_test_rccl_exchange runs the exchange's posting loop on a caller-supplied table (a recording fake: no GPU, no librccl) -- the unit test of the RCCL
 * branch's ordering, sizes and error path (a failing send still closes the group). */
typedef struct CorbRcclFns {
    int (*group_start)(void); int (*group_end)(void);
    int (*send)(const void* buf, size_t count, int datatype, int peer, void* comm, void* stream);
    int (*recv)(void* buf, size_t count, int datatype, int peer, void* comm, void* stream);
} CorbRcclFns;
int corb_comm_test_rccl_exchange(const CorbRcclFns* fns, const CorbPushMsg* sends, int n_sends, const CorbPushMsg* recvs, int n_recvs);

/* Asynchronous push.  corb_map_push_setup (collective, once): the root's stores' capacities / record sizes and its destination table travel to every rank, so that
 * afterwards EVERY rank can evaluate corb_map_push_plan itself.  corb_map_push_begin (collective): ONE all-gather of the headers (the only host synchronisation),
 * the same verdict on every rank without a second round, the records packed into the communicator's own staging buffers (grown on demand, never freed per push)
 * and the messages ENQUEUED on the communicator's stream; it returns without waiting for the transfer, which overlaps whatever the caller does next (tracking
 * runs on other streams).  The records of the pushed slots and the root's destination slots must not be touched until corb_map_push_wait (collective-free: every rank
 * waits for its own event) has returned; it finishes the root's bookkeeping (recv counts, stale host copies).  One push in flight per communicator.
 * Exclusion contract: every push locks a rank's stores while its records are packed and posted (a blocking RCCL push: until they have arrived; the in-process
 * transport: the receiving rank also while it copies them in), so a concurrent corb_ba_solve_store / corb_rebase_map_store / corb_track_* call on the same stores
 * sees records either before or after those phases.  Between corb_map_push_begin and corb_map_push_wait NO lock is held -- the transfer is in flight on the device:
 * a call that reads the root's destination slots in that window can see half-received records, and one that rewrites pushed slots can corrupt the message.  Keeping
 * the stores still in that window is the caller's job, as with the buffers of any asynchronous collective. */
int corb_map_push_setup(CorbComm* c, int root, CorbKfStore* kf, CorbMpStore* mp, const int32_t* kf_dst_first /* root: [world] */, const int32_t* mp_dst_first /* root: [world] or NULL */);
int corb_map_push_begin(CorbComm* c, const CorbMapPush* push, int root);
int corb_map_push_wait(CorbComm* c);

/* Server side of a push, on records: MapFusion::insertServerMapToGlobleMap (S/src/MapFusion.cpp:622-658; also :64-66, :126-131 for late arrivals):
 * Tcw <- Tcw * To2n for the keyframe slots, p <- Rwc (p - tcw) for the map-point slots, in place in device memory. */
int corb_rebase_map_store(const float* To2n, CorbKfStore* kf, const int32_t* kf_slots, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp);

/* Optimizer::GlobalBundleAdjustemnt / BundleAdjustment (C/src/Optimizer.cc:43-270) on store records -- what the server rank runs after a push + re-basing
 * (S/src/GlobalOptimize.cpp:444): vertices = the non-bad keyframes kf_slots (fixed iff mnId == 1 or CORB_KF_FIXED, :84-98) and the non-bad map points
 * mp_slots (:106-121); edges = every observation (keyframe id, feature) of those points whose keyframe is among kf_slots and not bad (:123-196), stereo iff
 * mvuRight[feature] >= 0, information mvInvLevelSigma2[octave]; intrinsics per keyframe.  The graph is built on the device from the records (no host
 * flattening, no uploads); results are written back into the records as the reference does (:216-262): loop_kf == 0 -> Tcw / world_pos, else TcwGBA /
 * pos_gba and ba_global_for_kf = loop_kf.  result->poses (n_kf x 16) / points (n_mp x 3) are optional copies (NULL = none).
 * loop_kf == 0: the reference follows SetWorldPos with pMP->UpdateNormalAndDepth() (:254-256).  With options->scale_factor > 0 the write-back does the same on the
 * records -- normal, min_distance, max_distance from the observers that are vertices of this solve (a global BA holds every keyframe of the map: all of them), with
 * the poses the solve left and mvScaleFactors rebuilt from scale_factor as ORBextractor.cc:418-424 does -- so that tracking calls (corb_track_search_local_points'
 * isInFrustum) can run on the records at once.  With scale_factor == 0 the three fields are left alone and the caller refreshes them (the adapter's
 * ReadBackMapPoints calls UpdateNormalAndDepth on the object) and re-files the points. */
int corb_ba_solve_store(CorbKfStore* kf, const int32_t* kf_slots, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp,
                        int iterations, int robust, volatile int* stop_flag, uint64_t loop_kf, CorbBAResult* result, const CorbBAOptions* options);

/* void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Cache* pCache) (C/src/Optimizer.cc:487-838) on store records -- what LocalMapping
 * runs after every keyframe (C/src/LocalMapping.cc:79).  The caller selects the window as the reference does (:493-546) and passes it as slots:
 *   kf_slots[0 .. n_local)     lLocalKeyFrames: free unless mnId == 1 or CORB_KF_FIXED (:569)
 *   kf_slots[n_local .. n_kf)  lFixedCameras: fixed (:582)
 *   mp_slots                   lLocalMapPoints (fixed iff CORB_MP_FIXED, :620)
 * Edges = every observation of those points in one of those keyframes (:626-700), built on the device from the records as for corb_ba_solve_store.
 * stages = the optimize(5) / "Check inlier observations" / optimize(10) schedule as CorbBAStage entries (the same two stages the host-pointer form
 * corb_ba_solve_staged is given for LocalBundleAdjustment); stop_flag = pbStopFlag with the reference's semantics (raised before the call: nothing is touched).
 * Afterwards, on the records (:760-836):
 *   apply_erase != 0: vToErase -- for every outlier observation the keyframe record's map-point id of that feature <- CORB_NO_MAP_POINT
 *     (pKFi->EraseMapPointMatch) and the observation leaves the point's list (pMP->EraseObservation, C/src/MapPoint.cc:192-217: mpRefKF moves to the first
 *     remaining observation if it was that keyframe; nObs -- 2 per stereo, 1 per monocular observation, an observation whose keyframe is outside the problem
 *     counts 1, one in a CORB_KF_BAD keyframe of the problem keeps its weight (it has no edge) -- <= 2 => SetBadFlag (:255-269): CORB_MP_BAD, n_obs = 0, the matches in its remaining keyframes OF THE PROBLEM cleared)
 *   Tcw of the local keyframes that are not CORB_KF_FIXED; world_pos of the local points that are not CORB_MP_FIXED, followed by
 *     MapPoint::UpdateNormalAndDepth (:424-472) over the observations whose keyframes are in the problem (the reference's window holds every observer:
 *     :530-545), with mvScaleFactors rebuilt from scale_factor (= ORBextractor's scaleFactor, 1.2 in every reference yaml) as ORBextractor.cc:418-424 does.
 * result->poses (n_kf x 16) / points (n_mp x 3): optional copies of the estimates.  erase_pairs[k] = (index into kf_slots, index into mp_slots) of the
 * k-th outlier observation in edge order (points in mp_slots order, within a point mObservations order); *n_erase = their number (may exceed erase_cap:
 * only the first erase_cap are stored). */
int corb_local_ba_store(CorbKfStore* kf, const int32_t* kf_slots, int n_local, int n_kf, CorbMpStore* mp, const int32_t* mp_slots, int n_mp,
                        const CorbBAStage* stages, int n_stages, float scale_factor, int apply_erase, volatile int* stop_flag,
                        CorbBAResult* result, int32_t* erase_pairs, int erase_cap, int* n_erase, const CorbBAOptions* options);

/* ---- tracking-thread calls on device-resident records ----
 * The current and the last Frame are records of a keyframe store (features, mvuRight, per-feature MapPoint ids = mvpMapPoints, pose, mvInvLevelSigma2;
 * record flag bit 1 of a feature = mvbOutlier), the map is a map-point store.  Nothing but a pose, a count and (optionally) the match array crosses PCIe.
 * corb_mp_store_build_index: mnId -> slot table (device memory, kept by the store) over slots first .. first+n; rebuild after records were added or moved.
 * CORB_ERR_ARG if two of the slots hold the same id. */
int corb_mp_store_build_index(CorbMpStore* s, int first, int n);
/* features of the record in `slot` as the host knows them (-1: empty slot, or filled from a device-side count) */
int corb_kf_store_count(CorbKfStore* s, int slot);
typedef struct CorbTrackCamera {
    float fx, fy, cx, cy, bf, mb;                 /* Frame::fx ... mbf, mb */
    float min_x, max_x, min_y, max_y;             /* mnMinX ... mnMaxY */
    int32_t nlevels; float scale[16];             /* mvScaleFactors */
} CorbTrackCamera;
/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) (C/src/ORBmatcher.cc:1470-1614) with
 * CurrentFrame = record cur_slot, LastFrame = record last_slot of `frames`: the last frame's features that hold a non-bad MapPoint of `map` and are not
 * outliers are projected with Tcw (the current frame's predicted pose; Tlw = the last frame's pose, for the forward / backward test), matched like
 * corb_search_by_projection_frame (same kernels), and CurrentFrame.mvpMapPoints[f] = the matched MapPoint's id is written into the record.  match (optional,
 * n(cur_slot) entries) = index of the last-frame feature or -1; *n_matches = the return value. */
int corb_track_search_last_frame(CorbKfStore* frames, int cur_slot, int last_slot, CorbMpStore* map, const float* Tcw /* 16 */, const float* Tlw /* 16 */,
                                 const CorbTrackCamera* cam, float th, int mono, float nnratio, int check_orientation, int32_t* match, int* n_matches);
/* int Optimizer::PoseOptimization(Frame *pFrame) (C/src/Optimizer.cc:272-485) on record `slot`: one edge per feature whose MapPoint id resolves to a non-bad
 * record of `map` (stereo iff mvuRight >= 0, information mvInvLevelSigma2[octave] of the record), start pose Tcw_in, the four rounds of the reference in ONE
 * kernel (as corb_pose_optimization_batch); mvbOutlier goes to the record's feature flags, the pose into the record (and Tcw_out); returns nInitialCorrespondences-nBad
 * in *n_inliers.  outlier (optional, n(slot) entries) = mvbOutlier.  discard_outliers: the "Discard outliers" loop of TrackWithMotionModel / TrackReferenceKeyFrame
 * (C/src/Tracking.cc:795-814, 919-940) on the record: an outlier feature loses its MapPoint and its mvbOutlier flag, and the point counts as seen in this frame
 * (mnLastFrameSeen: corb_track_search_local_points skips it); `outlier` still reports which features were rejected.  A pose optimisation with discard_outliers = 0
 * leaves mvbOutlier set, which the next frame's corb_track_search_last_frame honours (TrackLocalMap, Tracking.cc:1063-1083 keeps them too). */
int corb_track_pose_optimization(CorbKfStore* frames, int slot, CorbMpStore* map, const CorbTrackCamera* cam, const float* Tcw_in /* 16 */, float* Tcw_out /* 16 */,
                                 int discard_outliers, uint8_t* outlier, int32_t* n_inliers);
/* void Tracking::SearchLocalPoints() (C/src/Tracking.cc:1168-1216) on record `slot` with Tcw = the frame's current pose: bad MapPoints leave the frame; the
 * local MapPoints local_ids (mvpLocalMapPoints as ids; unknown ids and bad points are skipped) that the frame does not hold yet go through
 * Frame::isInFrustum(pMP, 0.5) (C/src/Frame.cc:270-329, MapPoint::PredictScale C/src/MapPoint.cc:500-514 with log_scale_factor = mfLogScaleFactor) on the
 * device, then ORBmatcher(nnratio).SearchByProjection(Frame&, vpMapPoints, th) (ORBmatcher.cc:45-131; the kernels of corb_search_by_projection_map) and
 * F.mvpMapPoints[f] = the matched id is written into the record.  match (optional, n(slot) entries) = index into local_ids or -1; tracked (optional,
 * n_local entries) = what isInFrustum left in the MapPoints (mbTrackInView -> valid); *n_in_view = nToMatch. */
int corb_track_search_local_points(CorbKfStore* frames, int slot, CorbMpStore* map, const uint64_t* local_ids, int n_local, const CorbTrackCamera* cam,
                                   const float* Tcw /* 16 */, float log_scale_factor, float th, float nnratio, int32_t* match, CorbTrackedPoint* tracked,
                                   int* n_matches, int* n_in_view);

/* int ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th) (C/src/ORBmatcher.cc:960-1116) on records -- what
 * LocalMapping::SearchInNeighbors runs per neighbour keyframe (C/src/LocalMapping.cc:475-560).  pKF = record `slot` of `kf` (features, mvuRight, descriptors;
 * Tcw = pKF->GetPose(), the camera centre is derived from it; cam = intrinsics, image bounds, mvScaleFactors, from which mvInvLevelSigma2 follows as
 * ORBextractor.cc:418-430 builds it); vpMapPoints = the records mp_slots of `map`.  A point takes part unless it is bad or already observed by pKF
 * (pMP->IsInKeyFrame(pKF), :990-993: its observation list holds the keyframe's id).  The search is corb_fuse's (same kernels, sim3 = 0):
 * best_idx[i] = the feature point i is fused into or -1, best_dist[i], *n_fused = the return value.
 * action[i] (optional): 0 not fused; 1 the feature held no MapPoint (:1097-1101) and point i is the first of vpMapPoints fused into it -- with apply != 0 the
 * records are updated as the reference does: pMP->AddObservation(pKF, bestIdx) (the observation enters the point's list in ascending keyframe id) and
 * pKF->AddMapPoint(pMP, bestIdx) (the keyframe record's map-point id of that feature); 2 the feature holds a MapPoint -- from before the call, or an earlier
 * point of this call -- i.e. the reference's MapPoint::Replace of the one with fewer observations (:1085-1096), which re-links whole observation lists:
 * left to the caller, in ascending i; 3 as 1, but the point's record has no room for another observation: nothing written, CORB_ERR_CAPACITY returned. */
int corb_fuse_store(CorbKfStore* kf, int slot, CorbMpStore* map, const int32_t* mp_slots, int n_points, const CorbTrackCamera* cam,
                    const float* Tcw /* 16 */, float log_scale_factor, float th, int apply, int32_t* best_idx, int32_t* best_dist, uint8_t* action, int* n_fused);

/* The part of MapPoint's serialised state (C/include/MapPoint.h:52-72) the 112-byte CorbMapPointRecord does not carry -- mnVisible, mnFound, mpReplaced -- lives in the 16
 * spare bytes of a record's 128-byte header on the device: zero after corb_mp_store_put, moved with the record by a push.  replaced_by: 0 = mpReplaced is NULL, else id + 1. */
typedef struct CorbMapPointCounters { int32_t n_visible, n_found; uint64_t replaced_by; } CorbMapPointCounters;
int corb_mp_store_set_counters(CorbMpStore* s, int first, int n, const CorbMapPointCounters* counters);
int corb_mp_store_get_counters(CorbMpStore* s, int first, int n, CorbMapPointCounters* counters);
/* The tracking / local-mapping / loop-closing scratch of MapPoint's serialised state (C/include/MapPoint.h:52-72) that neither the 112-byte header nor the counters carry:
 * with it a record holds EVERY field the reference's boost archive moves, so a push is a faithful stand-in for the archive.  It lives behind the observation lists of the
 * record (corb_mp_store_record_bytes covers it), is zero after corb_mp_store_put and travels with the record in corb_map_push_ex; the kernels of this library do not read it. */
typedef struct CorbMapPointScratch {
    int64_t first_kf_id, first_frame;           /* mnFirstKFid, mnFirstFrame */
    uint64_t track_reference_for_frame, last_frame_seen;      /* mnTrackReferenceForFrame, mnLastFrameSeen */
    uint64_t ba_local_for_kf, fuse_candidate_for_kf, loop_point_for_kf, corrected_by_kf, corrected_reference;      /* mnBALocalForKF ... mnCorrectedReference */
    float track_proj_x, track_proj_y, track_proj_xr, track_view_cos;      /* mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos */
    int32_t track_scale_level;                  /* mnTrackScaleLevel */
    int32_t n_obs_weight;                       /* nObs (2 per stereo observation, 1 per monocular one) */
    uint8_t track_in_view; uint8_t pad[7];      /* mbTrackInView */
} CorbMapPointScratch;                          /* 104 bytes */
int corb_mp_store_set_scratch(CorbMpStore* s, int first, int n, const CorbMapPointScratch* scratch);
int corb_mp_store_get_scratch(CorbMpStore* s, int first, int n, CorbMapPointScratch* scratch);
/* void MapPoint::Replace(MapPoint* pMP) (C/src/MapPoint.cc:277-316) on records: this = record slot_this, pMP = record slot_into of `map`; the keyframes are looked up by id among
 * the slots [kf_first, kf_first + kf_n) of `kf` (an observing keyframe that is not among them keeps its record; the observation lists are re-linked regardless).
 * As the reference: nothing if both are the same point; this loses its observations, becomes bad, mpReplaced = pMP; every observation (pKF, idx) of this, in list order, either
 * moves to pMP -- pKF->ReplaceMapPointMatch(idx, pMP) in the keyframe's record, pMP->AddObservation(pKF, idx) at its place in the ascending list -- or, when pMP is in that
 * keyframe already, is erased there (pKF->EraseMapPointMatch(idx)); pMP->IncreaseFound / IncreaseVisible by this' counters; pMP->ComputeDistinctiveDescriptors() over its
 * observations in non-bad keyframes of the named slots (same selection as corb_distinctive_descriptors).  *status: 0 done, 1 the same point.  CORB_ERR_CAPACITY (nothing
 * written) if pMP's observation list has no room.  What corb_fuse_store reports as action 2 is resolved with this call, point by point, in ascending order. */
int corb_mp_store_replace(CorbMpStore* map, int slot_this, int slot_into, CorbKfStore* kf, int kf_first, int kf_n, int* status);

/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist) (C/src/ORBmatcher.cc:1616-1744) on records --
 * the projection step of Tracking::Relocalization (C/src/Tracking.cc:1440-1500), which passes the MapPoints the frame holds as sAlreadyFound.  CurrentFrame = record
 * cur_slot of `frames` (Tcw = its pose estimate), pKF = record kf_slot of `kfs` (the same store or another one on the same device): pKF's features whose MapPoint id
 * resolves to a non-bad record of `map` and is not held by the frame are projected and matched like corb_search_by_projection_reloc (same kernels: closed image test,
 * octaves level-1 .. level+1, a frame feature that holds a MapPoint is skipped, rotation histogram when check_orientation); CurrentFrame.mvpMapPoints[f] = the matched id is
 * written into the frame's record.  match (optional, n(cur_slot) entries) = pKF feature index or -1; *n_matches = the return value. */
int corb_track_search_reloc(CorbKfStore* frames, int cur_slot, CorbKfStore* kfs, int kf_slot, CorbMpStore* map, const CorbTrackCamera* cam,
                            const float* Tcw /* 16 */, float log_scale_factor, float th, int orb_dist, int check_orientation, int32_t* match, int* n_matches);
/* int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th) (C/src/ORBmatcher.cc:425-538) on
 * records -- the projection search of a loop / map-fusion event right before CorrectLoop and the global BA (C/src/LoopClosing.cc:377; S/src/GlobalOptimize.cpp:199).
 * pKF = record `slot` of `kf`; vpPoints = the records mp_slots of `map` (the loop keyframe's covisible points); vpMatched = matched_ids, n(slot) MapPoint ids
 * (CORB_NO_MAP_POINT = NULL), read and written: a point takes part unless it is bad or its id is among matched_ids on entry, a feature that holds an id is skipped, and
 * matched_ids[idx] = the id of the point matched to feature idx.  Same kernels as corb_search_by_projection_scw.  match (optional, n(slot) entries) = index into
 * mp_slots or -1 (this call's matches only); *n_matches = the return value. */
int corb_search_by_projection_scw_store(CorbKfStore* kf, int slot, CorbMpStore* map, const int32_t* mp_slots, int n_points, const CorbTrackCamera* cam,
                                        const float* Scw /* 16 */, float log_scale_factor, float th, uint64_t* matched_ids, int32_t* match, int* n_matches);
/* int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, s12, R12, t12, th) (C/src/ORBmatcher.cc:1244-1468) on records -- LoopClosing::ComputeSim3's
 * guided search (C/src/LoopClosing.cc:318-330).  pKF1 / pKF2 = records slot1 / slot2 of `kf` (T1w / T2w = their poses; cam = pKF1's intrinsics, used for both directions as
 * the reference does); vpMatches12 on entry = matched12_ids (MapPoint ids per feature of KF1, CORB_NO_MAP_POINT = none; NULL = none at all): such a feature is skipped, and
 * so is the feature of KF2 that observes the matched point (GetIndexInKeyFrame through the point's observation list).  Same kernels as corb_search_by_sim3 in both
 * directions, agreement check on the host.  match12[i1] = feature of KF2 or -1 (new matches only, as in the host-array call); match12_ids (optional) = the MapPoint id
 * that feature holds, i.e. the new vpMatches12[i1]; *n_found = the return value. */
int corb_search_by_sim3_store(CorbKfStore* kf, int slot1, int slot2, CorbMpStore* map, const CorbTrackCamera* cam, float log_scale_factor,
                              const float* T1w /* 16 */, const float* T2w /* 16 */, const uint64_t* matched12_ids, float s12, const float* R12 /* 9 */, const float* t12 /* 3 */, float th,
                              int32_t* match12, uint64_t* match12_ids, int* n_found);

#ifdef __cplusplus
}
#endif
#endif
