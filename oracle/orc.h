/* CPU ORACLE for the CORB-SLAM hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's CPU algorithms (file:line citations are relative to
 * /root/reference/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline.  The product (libcorb_accel.so)
 * never links, loads or falls back to anything in oracle/.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and delegates its
 * pixel arithmetic to OpenCV (>=2.4.3, not vendored, not installable here) and its linear algebra
 * to Eigen3 (absent).  The third-party semantics restated here are those of OpenCV 2.4.8's scalar
 * (non-SIMD) C paths (the distro version of the reference's "ubuntu 14.04 / ROS indigo" target):
 *   cvRound            = lrint (round-half-to-even)
 *   cv::resize 8U INTER_LINEAR  : 11-bit fixed point, two passes (orc_resize_linear_u8)
 *   cv::GaussianBlur 7x7 s=2 8U : 8-bit kernel [18,34,49,55,49,34,18] per axis, (sum+2^15)>>16
 *   cv::FAST(9/16, nms)         : corner test + cornerScore<16> + strict 8-neighbour NMS
 *   cv::fastAtan2               : 7th-order odd polynomial, float, non-fused
 * Reference-undefined behaviour that this oracle DEFINES (documented in DESIGN.md):
 *   - quadtree size-tie order (reference: heap pointer order)   -> node creation order
 *   - Frame::mb read before initialisation in ComputeStereoMatches -> mb = mbf/fx
 *   - libm cosf/sinf (platform dependent)  -> orc_sincosf (double Cody-Waite + Taylor, rounded)
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* identical in layout to cv::KeyPoint (28 bytes) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} OrcKeyPoint;

typedef struct OrcExtractor OrcExtractor;

/* ---- extraction (corbslam_client/src/ORBextractor.cc) ---- */
OrcExtractor* orc_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void orc_orb_destroy(OrcExtractor* ex);
/* operator() : returns number of keypoints (<= cap) or -1 if cap too small */
int orc_orb_extract(OrcExtractor* ex, const uint8_t* img, int w, int h, int stride,
                    OrcKeyPoint* kps, uint8_t* desc, int cap);
/* tables */
void orc_orb_tables(const OrcExtractor* ex, float* scale, float* inv_scale, float* sigma2,
                    float* inv_sigma2, int* quota, int* umax16);
/* state left behind by the last orc_orb_extract (for per-stage parity checks) */
int orc_orb_level_dims(const OrcExtractor* ex, int level, int* w, int* h);
const uint8_t* orc_orb_level_data(const OrcExtractor* ex, int level);    /* pitch == w */
const uint8_t* orc_orb_blur_data(const OrcExtractor* ex, int level);     /* pitch == w, NULL if level had no kps */
int orc_orb_level_candidates(const OrcExtractor* ex, int level, OrcKeyPoint* out, int cap); /* pre-quadtree list */
int orc_orb_level_count(const OrcExtractor* ex, int level);              /* post-quadtree count */

/* stand-alone pieces (each follows one OpenCV / reference routine) */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride);
void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
int  orc_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, int nms,
                  OrcKeyPoint* out, int cap);
float orc_fast_atan2(float y, float x);
void  orc_sincosf(float x, float* s, float* c);
float orc_ic_angle(const uint8_t* img, int stride, int cx, int cy);
int orc_distribute_octree(const OrcKeyPoint* in, int n_in, int minX, int maxX, int minY, int maxY,
                          int N, OrcKeyPoint* out, int cap);

/* ---- matching (corbslam_client/src/ORBmatcher.cc, Frame.cc) ---- */
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);

typedef struct {
    float bf;       /* Frame::mbf */
    float mb;       /* Frame::mb as it SHOULD be at the call: mbf/fx (see header note) */
    int   nlevels;
    const float* scale;      /* mvScaleFactors  */
    const float* inv_scale;  /* mvInvScaleFactors */
} OrcStereoParams;

/* Frame::ComputeStereoMatches (Frame.cc:470-644).  Pyramids are those left in the extractors. */
int orc_stereo_match(const OrcExtractor* left, const OrcExtractor* right,
                     const OrcKeyPoint* kl, const uint8_t* dl, int nl,
                     const OrcKeyPoint* kr, const uint8_t* dr, int nr,
                     const OrcStereoParams* p, float* u_right, float* depth);

/* FeatureVector in flat form: node ids ascending, CSR offsets into idx */
typedef struct {
    int n_nodes;
    const uint32_t* node_id;
    const int32_t* offset;   /* n_nodes+1 */
    const uint32_t* idx;
} OrcFeatVec;

/* SearchByBoW(KeyFrame*,Frame&) (ORBmatcher.cc:162-291) and SearchByBoWInServer (294-423):
 * match_f[iF] = index of KF feature whose MapPoint was assigned, or -1.  variant 0.
 * SearchByBoW(KeyFrame*,KeyFrame*) (657-790): match12[i1] = idx2 or -1.  variant 1
 * (needs valid2, strict '<' TH_LOW, vbMatched2). */
int orc_search_by_bow(int variant,
                      const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const OrcFeatVec* fv1,
                      const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const OrcFeatVec* fv2,
                      float nnratio, int check_ori, int32_t* match_out);

typedef struct {
    float F12[9];            /* row-major 3x3 float */
    float ex, ey;            /* epipole in image 2 (ORBmatcher.cc:803-808) */
    int   nlevels;
    const float* scale2;     /* pKF2->mvScaleFactors */
    const float* sigma2_2;   /* pKF2->mvLevelSigma2  */
} OrcTriParams;

/* SearchForTriangulation (ORBmatcher.cc:792-958).  has_mp*: feature already has a MapPoint.
 * pairs_out: (idx1, idx2) sorted by idx1; returns count. */
int orc_search_for_triangulation(
        const uint8_t* desc1, const OrcKeyPoint* kp1, const float* uright1, const uint8_t* has_mp1, int n1, const OrcFeatVec* fv1,
        const uint8_t* desc2, const OrcKeyPoint* kp2, const float* uright2, const uint8_t* has_mp2, int n2, const OrcFeatVec* fv2,
        const OrcTriParams* p, int only_stereo, int check_ori, int32_t* pairs_out);

/* ---- projection-guided matchers (ORBmatcher.cc:45-131, 1470-1614; Frame.cc:230-245, 331-395) ---- */
typedef struct {                 /* what the matchers read from the current Frame */
    const OrcKeyPoint* keys_un; const float* u_right; const uint8_t* desc; int n;
    const uint8_t* claimed;      /* feature already holds a MapPoint with Observations()>0 */
    float min_x, min_y, max_x, max_y;     /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    const float* scale; int nlevels;      /* mvScaleFactors */
} OrcFrameView;
typedef struct {                 /* MapPoint tracking fields set by Frame::isInFrustum */
    float proj_x, proj_y, proj_xr, view_cos;   /* mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos */
    int32_t level;               /* mnTrackScaleLevel */
    uint8_t valid;               /* mbTrackInView && !isBad() */
    uint8_t claims;              /* Observations()>0: the assignment blocks later map points */
    uint8_t pad[2];
} OrcTrackedPoint;
typedef struct {                 /* one feature of the last frame */
    float world[3];              /* pMP->GetWorldPos() */
    float angle;                 /* LastFrame.mvKeysUn[i].angle */
    int32_t octave;              /* LastFrame.mvKeys[i].octave */
    uint8_t valid;               /* has a MapPoint and !mvbOutlier[i] */
    uint8_t claims;              /* pMP->Observations()>0 */
    uint8_t pad[2];
} OrcLastPoint;
/* match[i] (per current-frame feature) = index of the assigned map point / last-frame feature, or -1; returns nmatches */
int orc_search_by_projection_map(const OrcFrameView* F, const OrcTrackedPoint* mp, const uint8_t* mp_desc, int n_mp,
                                 float th, float nnratio, int32_t* match);
int orc_search_by_projection_frame(const OrcFrameView* C, const float* Tcw, const float* Tlw, float fx, float fy, float cx, float cy,
                                   float bf, float mb, const OrcLastPoint* last, const uint8_t* last_desc, int n_last,
                                   float th, int bMono, int check_ori, int32_t* match);

/* ---- projection matchers against a KeyFrame / for relocalisation (ORBmatcher.cc:960-1468, 1616-1744; KeyFrame.cc:700-735) ---- */
typedef struct {                 /* the target of the projection: a KeyFrame (or, for relocalisation, the current Frame) */
    const OrcKeyPoint* keys_un; const float* u_right; const uint8_t* desc; int n;
    float min_x, min_y, max_x, max_y;     /* mnMinX .. mnMaxY */
    const float* scale; const float* inv_level_sigma2; int nlevels;   /* mvScaleFactors, mvInvLevelSigma2 */
    float log_scale_factor;      /* mfLogScaleFactor */
    float fx, fy, cx, cy, bf;
} OrcKeyFrameView;
typedef struct {                 /* one MapPoint as the matchers read it */
    float world[3];              /* GetWorldPos() */
    float normal[3];             /* GetNormal() (Fuse only) */
    float min_distance, max_distance;     /* mfMinDistance, mfMaxDistance (the Invariance getters scale them by 0.8f / 1.2f) */
    float angle;                 /* relocalisation: pKF->mvKeysUn[i].angle of the keyframe feature that holds the point */
    uint8_t valid;               /* the skip tests evaluated by the adapter: non-NULL, !isBad(), not already found / in the keyframe / matched */
    uint8_t pad[3];
} OrcMapPointView;
/* SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (:1616-1744): match[iFeature] = map point index or -1 */
int orc_search_by_projection_reloc(const OrcKeyFrameView* C, const uint8_t* claimed, const float* Tcw, const OrcMapPointView* pts,
                                   const uint8_t* desc, int n, float th, int orb_dist, int check_ori, int32_t* match);
/* Fuse(KeyFrame*, vpMapPoints, th) (:960-1116; sim3 = 0, T = Tcw, Ow = GetCameraCenter()) and
 * Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (:1118-1241; sim3 = 1, T = Scw, Ow ignored).
 * best_idx[i] = keyframe feature the point would be fused into (bestDist <= TH_LOW) or -1; returns their number (nFused). */
int orc_fuse(const OrcKeyFrameView* K, const float* T, const float* Ow, int sim3, const OrcMapPointView* pts, const uint8_t* desc, int n,
             float th, int32_t* best_idx, int32_t* best_dist);
/* SearchForInitialization(Frame& F1, Frame& F2, vbPrevMatched, vnMatches12, windowSize) (:540-655): prev_matched n1 x 2 floats in / out; match12[i1] = feature of F2 or -1 */
int orc_search_for_initialization(const OrcFrameView* F1, const OrcFrameView* F2, float* prev_matched, int window_size, float nnratio, int check_ori, int32_t* match12);
/* SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (:425-538): claimed[idx] = vpMatched[idx] != NULL on entry; match[idx] = index of the point this
 * call writes into vpMatched[idx] or -1; returns nmatches */
int orc_search_by_projection_scw(const OrcKeyFrameView* K, const uint8_t* claimed, const float* Scw, const OrcMapPointView* pts, const uint8_t* desc, int n,
                                 float th, int32_t* match);
/* SearchBySim3 (:1244-1468): match12[i1] = feature of KF2 or -1 (mutually consistent matches only); returns nFound */
int orc_search_by_sim3(const OrcKeyFrameView* K1, const OrcKeyFrameView* K2, const float* T1w, const float* T2w,
                       const OrcMapPointView* pts1, const uint8_t* desc1, const OrcMapPointView* pts2, const uint8_t* desc2,
                       float s12, const float* R12, const float* t12, float th, int32_t* match12);

/* ---- map maintenance next to the hot path (MapPoint.cc:337-402, S/src/MapFusion.cpp:622-658) ---- */
void orc_distinctive_descriptors(const uint8_t* desc, const int32_t* offset, int n_points, int32_t* best_idx);
/* MapPoint::Replace (MapPoint.cc:277-316) on flat observation lists: see orc_map.c */
int orc_mappoint_replace(uint64_t id_this, uint64_t id_into, const uint64_t* kf_this, const uint32_t* idx_this, int n_this,
                         uint64_t* kf_into, uint32_t* idx_into, int32_t* n_into, int cap_into, uint8_t* action, const int32_t* counters_this, int32_t* counters_into);
void orc_rebase_map(const float* To2n, float* poses, int n_poses, float* points, int n_points);

/* ---- global bundle adjustment (Optimizer.cc:43-270 + g2o) ---- */
typedef struct {
    int32_t pose, point;     /* indices into the pose / point arrays */
    float u, v, ur;          /* ur < 0  => monocular edge */
    float inv_sigma2;
} OrcBAEdge;

typedef struct {
    int n_poses, n_points, n_edges;
    const float* poses;          /* n_poses x 16, row-major 4x4 Tcw (float, as cv::Mat) */
    const uint8_t* pose_fixed;
    const float* points;         /* n_points x 3 */
    const uint8_t* point_fixed;
    const OrcBAEdge* edges;
    float fx, fy, cx, cy, bf;    /* shared camera (used when intr == NULL) */
    const float* intr;           /* n_poses x 5: fx, fy, cx, cy, bf of every keyframe (e->fx = pKF->fx ..., Optimizer.cc:160-163, 189-193), or NULL */
} OrcBAProblem;

typedef struct {
    float* poses;                /* n_poses x 16 out */
    float* points;               /* n_points x 3 out */
    double* chi2;                /* [iters+1]: chi2 before iteration 0, then after each outer iteration */
    double* lambda;              /* [iters] lambda after each outer iteration */
    int iters_done;
    int trials_total;
} OrcBAResult;

int orc_ba_solve(const OrcBAProblem* p, int iters, int robust, volatile int* stop, OrcBAResult* r);
/* reduced-system solver of the oracle: 0 = auto (dense LDL^T up to 256 free poses, block-sparse LDL^T above), 1 = dense, 2 = block-sparse
 * (the reference's solver class: SimplicialLDLT on the sparse Schur complement, linear_solver_eigen.h:94-232) */
void orc_ba_set_solver(int solver);

/* one optimize() call + the outlier test that follows it (LocalBundleAdjustment / PoseOptimization) */
typedef struct {
    int iterations, robust;
    float chi2_mono, chi2_stereo;   /* 5.991 / 7.815 */
    int check_depth;                /* also test isDepthPositive() (LocalBundleAdjustment) */
    int recompute_inactive;         /* computeError() on inactive edges before the test (PoseOptimization) */
    int allow_reactivate;           /* inactive edges may become active again (PoseOptimization) */
    int reset_estimates;            /* restart from the input estimate (PoseOptimization) */
    int float_compare;              /* compare chi2 as float (PoseOptimization) */
    float huber_mono, huber_stereo; /* Huber deltas: (float)sqrt(5.991), (float)sqrt(7.815) (Optimizer.cc:309-310, 601-602) */
} OrcBAStage;
int orc_ba_solve_staged(const OrcBAProblem* p, const OrcBAStage* stages, int n_stages, volatile int* stop, OrcBAResult* r, uint8_t* edge_outlier);

/* ---- Optimizer::OptimizeSim3 (Optimizer.cc:1119-1311) ---- */
typedef struct {
    int n;                              /* correspondences (pairs of edges e12 / e21) */
    const float* p1c; const float* p2c; /* n x 3: P3D1c = R1w*P3D1w + t1w, P3D2c likewise (Converter::toVector3d of the float Mats) */
    const float* obs1; const float* obs2;          /* n x 2: pKF1->mvKeysUn[i].pt, pKF2->mvKeysUn[i2].pt */
    const float* inv_sigma2_1; const float* inv_sigma2_2;
    float fx1, fy1, cx1, cy1, fx2, fy2, cx2, cy2;  /* K1, K2 */
} OrcSim3Problem;
/* g2oS12 in/out as (R 3x3 row-major, t, s) in double; removed[i] = 1 when vpMatches1[idx] is nulled; returns nIn (0 and S12 unchanged when
 * fewer than 10 correspondences survive the first round) */
int orc_optimize_sim3(const OrcSim3Problem* p, double* R12, double* t12, double* s12, float th2, int fix_scale, uint8_t* removed,
                      int* iters_done, int* trials);

/* ---- Optimizer::OptimizeEssentialGraph (Optimizer.cc:840-1117) ---- */
int orc_optimize_essential_graph(int K, double* S /* K x 8: quaternion xyzw, t, s */, const uint8_t* fixed, int E, const int32_t* vi, const int32_t* vj,
                                 const double* meas /* E x 8 */, int iters, int fix_scale, double* chi2_hist, int* iters_done, int* trials_done);
void orc_essential_graph_apply(int K, const double* S_old, const double* S_new, float* Tiw_out, int M, const int32_t* ref, float* points);

#ifdef __cplusplus
}
#endif
#endif
