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

struct CorbProjPose { float Tcw[16]; float fx, fy, cx, cy, bf; int forward, backward; };

struct CorbProjDev {
    int n, nq;                        // features of the current frame, queries
    float min_x, min_y, max_x, max_y, winv, hinv;
    float scale[CORB_MAX_LEVELS];
    float nnratio; int ratio_test, check_ori;
    const CorbKeyPoint* keys; const float* u_right; const unsigned long long* desc; const unsigned char* claimed;
    const unsigned long long* qdesc;  // [nq][4]
    CorbProjQuery* query;
    int* feat_cell; int* cell_off; int* cell_idx;
    unsigned long long* cand_key; unsigned char* cand_oct; int* cand_cnt;
    int* ev_feat; int* ev_bin;
    int* match; int* n_matches; int* status;
};

void corb_launch_projection(const CorbProjDev& d, const CorbTrackedPoint* mp, const CorbLastPoint* last, const CorbProjPose* pose, float th, hipStream_t s);
