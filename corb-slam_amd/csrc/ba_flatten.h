// ba_flatten.h -- argument block of the device-side graph flattening (ba_flatten.hip, driven by corb_ba.cpp: corb_ba_solve_device)
#pragma once
#include "corb_internal.h"

#define FLAT_MAXLIST 0      // scal[]: longest keyframe edge list
#define FLAT_MAXROW  2      //         most blocks in one block row
#define FLAT_PAIRS   1      //         sum over the free landmarks of (free-keyframe observations)^2: an upper bound of the Schur pair lists' length (local windows)
#define FLAT_NSCAL   4

struct BAFlattenDev {
    // the problem (device arrays; edges grouped by point)
    int K, M, E;
    float* poses; const uint8_t* pose_fixed; float* points; const uint8_t* point_fixed; const CorbBAEdge* edges; const float* intr; const int* edge_off;
    // per point / per pose scratch
    int *lflag, *cntA, *cntB, *nfree_pt;      // [M]
    int *lidx, *eoffA, *eoffB;                // [M + 1] exclusive scans
    int *pflag, *pidx;                        // [K], [K + 1]
    uint8_t* pt_touched;                      // [M]
    int *pcnt, *pcur;                         // [nP] edges per free pose, fill cursors
    int* erel;                                // [nE] maps: an edge's place in its keyframe's unordered list (flat_pose_count_kernel); NULL: counts by flat_edge_kernel
    int *rowcnt, *ucnt, *ubase;               // [nP] blocks per row, blocks on / above the diagonal, [nP + 1] their scan
    int* scal;                                // [FLAT_NSCAL]
    // outputs: the arrays of BAFlat
    int *e_pose, *e_point, *e_vpose, *e_vpoint, *loff, *lnfree, *poff, *pedge, *pose_vertex, *point_vertex, *plm;
    double *e_obs, *e_w, *cam, *state; unsigned char* e_dim;
    int *bsr_rowptr, *bsr_col, *bsr_diag, *uinfo;
    int* e_src;                               // (optional) [nE] the problem's edge behind every flattened edge
};
void flat_launch_points(const BAFlattenDev& d, hipStream_t s);
void flat_launch_state_in(const BAFlattenDev& d, hipStream_t s);
void flat_launch_state_out(const BAFlattenDev& d, hipStream_t s);
void flat_launch_edges(const BAFlattenDev& d, hipStream_t s, int few_poses = 0);      // few_poses = nP when nP <= 64: the per-keyframe counts go through a workgroup's LDS first
void flat_launch_pose_lists(const BAFlattenDev& d, int nE, hipStream_t s);
void flat_launch_pose_count(const BAFlattenDev& d, int nE, hipStream_t s);      // d.erel != NULL: counts (d.pcnt), places (d.erel), the longest list (FLAT_MAXLIST)
void flat_launch_pose_fill(const BAFlattenDev& d, int nE, hipStream_t s);       // d.pedge from d.poff + d.erel (unordered per keyframe: flat_launch_pose_sort follows)
int flat_launch_pose_sort(const BAFlattenDev& d, int nP, int max_list, hipStream_t s);      // -1: a keyframe with more than 16 384 observations
void flat_launch_rows(const BAFlattenDev& d, int nP, bool fill, hipStream_t s);
// local windows (dense reduced system): the FULL block pattern -- every pair of free keyframes, nP^2 blocks, nP (nP + 1) / 2 on / above the diagonal: known without a
// look at the lists
// a few keyframes (local windows): a workgroup per keyframe compacts ITS edges out of the edge array in order -- the list comes out ascending, no atomics, no sort
void flat_launch_pose_lists_ordered(const BAFlattenDev& d, int nP, int nE, hipStream_t s);
void flat_launch_full_pattern(const BAFlattenDev& d, int nP, hipStream_t s);
// the flattening's counts as ONE block (one copy): out[0..5] = free landmarks, edges of free landmarks, edges of fixed landmarks, free keyframes, scal[FLAT_PAIRS],
// the problem's edge count (edge_off[M]); out[6] = *extra (the caller's status word, optional)
void flat_launch_counts(const BAFlattenDev& d, const int* extra, int* out, hipStream_t s);
// outlier[e_src[j]] = 1 for every flattened edge j that is not in the active set
void flat_launch_outliers(const unsigned char* active, const int* e_src, int nE, unsigned char* outlier, hipStream_t s);
// outlier[e] of the problem's edges between a fixed keyframe and a fixed map point (outside the flattened graph): the depth test of the classification stages that ran
void flat_launch_fixed_edge_outliers(const BAFlattenDev& d, const CorbBAStage* used, int n_used, unsigned char* outlier, hipStream_t s);
