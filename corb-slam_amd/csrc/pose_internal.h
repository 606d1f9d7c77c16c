// pose_internal.h -- device argument block of the fused single-pose Levenberg-Marquardt kernel
// (Optimizer::PoseOptimization, C/src/Optimizer.cc:272-485, and any staged problem with ONE free pose and no free point).
#pragma once
#include "corb_internal.h"
#include "corb_accel.h"

#define CORB_POSE_MAX_STAGES 8

struct CorbPoseDev {
    int n_problems;
    const int* edge_off;          // [n_problems + 1] edge range of each problem
    const double* pt;             // [E][3] world position of the (fixed) map point of every edge
    const double* obs;            // [E][3] u, v, u_right
    const double* w;              // [E]    information scale (invSigma2)
    const unsigned char* dim;     // [E]    2 mono / 3 stereo
    const double* cam;            // [n_problems][5] fx fy cx cy bf
    double* pose;                 // [n_problems][7] quaternion x y z w, translation -- in: start, out: result
    double* last_chi2;            // [E] chi2 of the edge's last computeError()
    unsigned char* active;        // [E] out: 1 = inlier after the last stage
    int* counters;                // [n_problems][4] iterations, trials, touched, inliers
    int n_stages;
    const int* stage_limit;       // [n_problems] stages this problem runs (NULL: n_stages) -- PoseOptimization stops after its first round when the graph has
                                  // fewer than 10 edges (`if(optimizer.edges().size()<10) break;`, Optimizer.cc:470-471)
    CorbBAStage stages[CORB_POSE_MAX_STAGES];
};

void pose_launch_optimize(const CorbPoseDev& d, int max_edges, hipStream_t s);      // max_edges: of one problem of the launch (host-known)
