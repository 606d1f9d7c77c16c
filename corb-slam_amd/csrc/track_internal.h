// track_internal.h -- tracking-thread calls on device-resident records (corb_track.cpp, track_kernels.hip): the current / last frame are keyframe-store
// records, the map is a map-point store; what the host-pointer calls receive as flat views is gathered from the records by kernels.
#pragma once
#include "proj_internal.h"
#include "pose_internal.h"
#include "store_internal.h"
#include "device_util.h"

#define CORB_FEATURE_HAS_MP   1u      // record flags[i] bit 0: "has a good MapPoint" (corb_kf_store_set_flags; SearchForTriangulation on slots)
#define CORB_FEATURE_OUTLIER  2u      // record flags[i] bit 1: mvbOutlier[i], written by corb_track_pose_optimization
#define CORB_FEATURE_DISCARDED 4u     // record flags[i] bit 2: the feature's MapPoint was discarded as an outlier (Tracking.cc:919-940): the feature holds no MapPoint any
                                      // more, but the id stays in the record as the point's mnLastFrameSeen == this frame (SearchLocalPoints skips it)

struct TrackDev {
    char* cur; const char* last; int F;                  // records of the current / last frame (RecLayout(F))
    const char* mp_base; size_t mp_bytes; CorbIdTable idt;
    CorbLastPoint* lastp; unsigned long long* qdesc;     // [n_last] views of the last frame's map points, their descriptors [n_last][4]
    unsigned char* claimed;                              // [n_cur]
    const int* match;                                    // [n_cur] result of the matcher
    int n_cur, n_last;
};
// lastp / qdesc <- the last frame's features with a usable MapPoint; claimed <- the current frame's features that already hold an observed MapPoint
void track_launch_prepare_last(const TrackDev& t, hipStream_t s);
// CurrentFrame.mvpMapPoints[f] = LastFrame.mvpMapPoints[match[f]] for the matched features (ORBmatcher.cc:1582 / 1597)
void track_launch_scatter_last(const TrackDev& t, hipStream_t s);

struct TrackPoseDev {
    char* cur; int F, n_cur;
    const char* mp_base; size_t mp_bytes; CorbIdTable idt;
    int* edge_off;                                       // [2] = {0, E}
    int* stage_limit;                                    // [1] rounds to run: 0 if E < 3 (`return 0`, Optimizer.cc:369-370), 1 if E < 10 (:470-471), else 4
    double* pt; double* obs; double* w; unsigned char* dim;     // the arrays of CorbPoseDev, E <= n_cur entries used
    int* efeat;                                          // [E] feature of edge e
    const unsigned char* active; const double* pose; const int* counters;     // results of the optimisation
    int discard;                                         // outliers lose their MapPoint (TrackWithMotionModel / TrackLocalMap's "Discard outliers")
    double* result;                                      // [10] what the host reads, as one block (one copy): pose[7] | counters[4] and edge_off[2] as six ints
};
// one edge per feature that holds a usable MapPoint, in feature order (Optimizer.cc:300-366); edge_off[1] = their number
void track_launch_pose_gather(const TrackPoseDev& t, hipStream_t s);
// mvbOutlier -> record flags, the optimised pose -> the record's Tcw (when the graph had an active edge: counters[2])
void track_launch_pose_finish(const TrackPoseDev& t, hipStream_t s);
// id table of a map-point store: slots [first, first + n)
void track_launch_index_store(const char* mp_base, size_t mp_bytes, int first, int n, CorbIdTable idt, int* dup, hipStream_t s);

struct TrackLocalDev {
    char* cur; int F, n_cur;
    const char* mp_base; size_t mp_bytes; CorbIdTable idt;       // the map
    CorbIdTable inframe;                                         // ids held by the frame's features (built per call)
    const unsigned long long* ids; int n_local;                  // mvpLocalMapPoints as ids
    CorbTrackedPoint* tracked; unsigned long long* qdesc;        // [n_local] mTrackProj*, mnTrackScaleLevel, mTrackViewCos; descriptors [n_local][4]
    unsigned char* claimed; const int* match; int* n_in_view;
    float Tcw[16], Ow[3], fx, fy, cx, cy, bf, min_x, max_x, min_y, max_y, log_scale, cos_limit; int nlevels;
};
// Tracking::SearchLocalPoints (Tracking.cc:1168-1204): bad MapPoints leave the frame, the frame's own MapPoints are excluded, Frame::isInFrustum(pMP, 0.5)
// (Frame.cc:270-329) for the rest; claimed <- the frame's features that hold an observed MapPoint
void track_launch_prepare_local(const TrackLocalDev& t, hipStream_t s);
// F.mvpMapPoints[f] = vpMapPoints[match[f]] (ORBmatcher.cc:122)
void track_launch_scatter_local(const TrackLocalDev& t, hipStream_t s);

// ---- LocalMapping::SearchInNeighbors' Fuse on records (corb_fuse_store) ----
struct FuseStoreDev {
    char* kf_rec; int F, n_feat;                                 // the target keyframe's record
    char* mp_base; size_t mp_bytes; int max_obs; const int* mp_slots; int n_points;
    CorbMapPointView* pts; unsigned long long* qdesc;            // [n_points] views of the candidate points, descriptors [n_points][4]
    const int* best_idx;                                         // [n_points] result of the search
    int* claim;                                                  // [n_feat] lowest point index that was fused into the feature
    unsigned char* action;                                       // [n_points] 0 none, 1 added, 2 the feature holds a MapPoint (Replace pending), 3 added but the observation list is full
    int apply;
};
// pts / qdesc <- the records: a point takes part unless it is bad or already observed by the keyframe (ORBmatcher.cc:987-993)
void fuse_launch_prepare(const FuseStoreDev& t, hipStream_t s);
// the map update of the fused points (ORBmatcher.cc:1083-1104), see corb_accel.h
void fuse_launch_apply(const FuseStoreDev& t, hipStream_t s);

// ---- relocalisation's SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) on records (corb_track_search_reloc) ----
struct RelocStoreDev {
    char* cur; int F_cur, n_cur;                                 // the current frame's record
    const char* kf; int F_kf, n_kf;                              // pKF's record
    const char* mp_base; size_t mp_bytes; CorbIdTable idt;       // the map
    CorbIdTable inframe;                                         // sAlreadyFound: the ids the frame's features hold (built per call)
    CorbMapPointView* pts; unsigned long long* qdesc;            // [n_kf] views of pKF's MapPoints, descriptors [n_kf][4]
    unsigned char* claimed; const int* match;                    // [n_cur]
};
void reloc_launch_prepare(const RelocStoreDev& t, hipStream_t s);
void reloc_launch_scatter(const RelocStoreDev& t, hipStream_t s);

// ---- ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) on records (corb_search_by_projection_scw_store) ----
struct ScwStoreDev {
    const char* kf_rec; int F, n_feat;                           // pKF's record
    const char* mp_base; size_t mp_bytes; const int* mp_slots; int n_points;      // vpPoints as slots of the map
    unsigned long long* matched;                                 // [n_feat] vpMatched as MapPoint ids (CORB_NO_MAP_POINT = NULL), in / out
    CorbIdTable found;                                           // spAlreadyFound: the ids of vpMatched on entry (built per call)
    CorbMapPointView* pts; unsigned long long* qdesc;            // [n_points] views of the candidate points, descriptors [n_points][4]
    unsigned char* claimed; const int* match;                    // [n_feat]
};
void scw_launch_prepare(const ScwStoreDev& t, hipStream_t s);
void scw_launch_scatter(const ScwStoreDev& t, hipStream_t s);

// ---- ORBmatcher::SearchBySim3 on records (corb_search_by_sim3_store) ----
struct Sim3StoreDev {
    const char* kf1; const char* kf2; int F, n1, n2;             // the two keyframes' records (one store)
    const char* mp_base; size_t mp_bytes; int max_obs; CorbIdTable idt;
    const unsigned long long* matched12;                         // [n1] vpMatches12 on entry as MapPoint ids (NULL = none)
    unsigned char* already1; unsigned char* already2;            // [n1], [n2] (already2 zeroed by the caller)
    CorbMapPointView* pts1; CorbMapPointView* pts2; unsigned long long* qdesc1; unsigned long long* qdesc2;
};
void sim3_launch_prepare(const Sim3StoreDev& t, hipStream_t s);
