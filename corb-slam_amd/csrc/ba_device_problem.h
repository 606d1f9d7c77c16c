// ba_device_problem.h -- a bundle-adjustment problem whose arrays live in device memory (corb_ba_store.cpp builds it from store records, corb_ba.cpp solves it)
#pragma once
#include "corb_internal.h"

struct CorbBADeviceProblem {          // CorbBAProblem with device pointers; per-keyframe intrinsics always present
    int n_poses, n_points, n_edges;
    float* poses;                     // [n_poses][16], updated in place (fixed / untouched vertices keep their values)
    const uint8_t* pose_fixed;
    float* points;                    // [n_points][3], updated in place
    const uint8_t* point_fixed;
    const CorbBAEdge* edges;          // grouped by point: the edges of point j are edges[edge_off[j] .. edge_off[j+1])
    const float* intr;                // [n_poses][5]
    const int* edge_off;              // [n_points + 1]
};
// corb_ba_solve_ex on device arrays.  result->poses / points are ignored (the estimates are written into p->poses / p->points on the device).
int corb_ba_solve_device(const CorbBADeviceProblem* p, int iterations, int robust, volatile int* stop_flag, CorbBAResult* result, int device, const CorbBAOptions* options);
// corb_ba_solve_staged on device arrays, for local windows (LocalBundleAdjustment on store records): the graph is flattened on the device (ba_flatten.hip) with ONE
// read-back of counts, the optimize() calls and the classifications between them run where the estimates are, and nothing else travels until the end.
// p->n_edges may be -1: the arrays were sized by a bound and the count (edge_off[n_points]) comes down with the flattening's counts -- *n_edges_out receives it, and
// *status_out the caller's device status word (status, optional), both as soon as the counts are there (also when the call declines).
// ready (optional): an event the problem's arrays are complete behind.
// outlier (device, one byte per edge): 1 = classified outlier after the last stage, in the problem's edge order.  *applicable = 0: the call declined BEFORE touching anything
// (a window for the one-workgroup optimiser, more than 64 free keyframes, a stage that restarts from the input estimates, ...): the caller takes the host route.
struct CorbBAStage;
int corb_ba_staged_device(const CorbBADeviceProblem* p, const CorbBAStage* stages, int n_stages, volatile int* stop_flag, CorbBAResult* result, uint8_t* outlier,
                          hipEvent_t ready, const int* status, int* n_edges_out, int* status_out, int device, const CorbBAOptions* options, int* applicable);
