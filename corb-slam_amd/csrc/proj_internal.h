// proj_internal.h -- device argument block of the projection-guided matchers.
#pragma once
#include "corb_internal.h"

#define PROJ_COLS 64                  // FRAME_GRID_COLS (C/include/Frame.h:39)
#define PROJ_ROWS 48                  // FRAME_GRID_ROWS (C/include/Frame.h:38)
#define PROJ_CELLS (PROJ_COLS * PROJ_ROWS)
#define PROJ_CAND_CAP 256             // candidates kept per query (overflow is reported, never silent)

struct CorbProjQuery {
    float x, y, r;                    // search window centre / half size (GetFeaturesInArea arguments)
    int min_level, max_level;
    float ur_ref;                     // projected right coordinate
    float angle;                      // keypoint angle of the query (rotation histogram)
    unsigned char valid, claims, pad[2];
};

// how a world point reaches the target image in the keyframe-target matchers (relocalisation projection, Fuse x2, SearchBySim3)
struct CorbProjTf {
    float A[12], B[12];               // affine maps [R|t] (row-major 3x4): p' = A p ; optionally p'' = B p' (Sim3 chains two)
    float Ow[3];                      // camera centre for the distance / viewing-angle tests
    float fx, fy, cx, cy, bf, log_scale, th;
    int two;                          // apply B after A
    int reloc;                        // SearchByProjection(Frame&, KeyFrame*): no depth test, closed image test, u = (fx*x)*invz + cx
    int invz_double;                  // invz = (float)(1.0 / z) instead of 1.0f / z
    int dist_from_cam;                // dist3D = |p''| (Sim3) instead of |p - Ow|
    int check_normal;                 // viewing angle test PO.Pn >= 0.5 dist3D (Fuse)
    int lvl_hi;                       // accepted octaves: [level-1, level+lvl_hi]
    int nlevels;
};

struct CorbProjPose { float Tcw[16]; float fx, fy, cx, cy, bf; int forward, backward; };

struct CorbProjDev {
    int n, nq;                        // features of the current frame, queries
    float min_x, min_y, max_x, max_y, winv, hinv;
    float scale[CORB_MAX_LEVELS];
    float nnratio; int ratio_test, check_ori;
    int check_uright;                 // candidates must agree with the projected right coordinate (Frame variants)
    int th_dist;                      // accepted best distance (TH_HIGH, ORBdist, TH_LOW)
    int chi2_check;                   // Fuse(KeyFrame*, vpMapPoints): reprojection chi2 test per candidate
    float inv_sigma2[CORB_MAX_LEVELS];
    const CorbKeyPoint* keys; const float* u_right; const unsigned long long* desc; const unsigned char* claimed;
    const unsigned long long* qdesc;  // [nq][4]
    CorbProjQuery* query;
    int* feat_cell; int* cell_off; int* cell_idx;
    unsigned long long* cand_key; unsigned char* cand_oct; int* cand_cnt;
    int* ev_feat; int* ev_bin;
    int* match; int* n_matches; int* status;
    int* best_idx; int* best_dist;    // [nq] independent best candidate per query (Fuse, SearchBySim3)
    int cand_cap;                     // candidates kept per query: 0 = PROJ_CAND_CAP (every matcher but SearchForInitialization, whose windows are 100 px wide)
};

void corb_launch_projection_points(const CorbProjDev& d, const CorbMapPointView* pts, const CorbProjTf& tf, int greedy, hipStream_t s);
void corb_launch_search_for_initialization(const CorbProjDev& d, const CorbKeyPoint* keys1, float* prev_matched, float window, hipStream_t s);
void corb_launch_projection(const CorbProjDev& d, const CorbTrackedPoint* mp, const CorbLastPoint* last, const CorbProjPose* pose, float th, hipStream_t s);
