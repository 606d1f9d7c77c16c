// ba_store_internal.h -- argument block of the kernels that build the BA graph from store records (store_kernels.hip, corb_ba_store.cpp)
#pragma once
#include "corb_internal.h"
#include "device_util.h"

#define BAS_DUPLICATE_KF 1      // a keyframe id occurs twice among kf_slots
#define BAS_BAD_FEATURE  2      // an observation's feature index lies outside its keyframe

struct BAStoreDev {
    int n_kf, n_mp, max_features, max_obs;
    int n_local;                        // vertices [n_local, n_kf) are fixed whatever their flags say (lFixedCameras of LocalBundleAdjustment); n_kf: none
    char* kf_base; size_t kf_bytes; const int* kf_slots;
    char* mp_base; size_t mp_bytes; const int* mp_slots;
    CorbIdTable tab;                    // keyframe id -> vertex index
    // the problem as device arrays (the layout of CorbBAProblem)
    float* poses; float* intr; uint8_t* pose_fixed; uint8_t* kf_bad;
    float* points; uint8_t* point_fixed; uint8_t* mp_bad;
    int* edge_cnt; int* edge_off;       // per map point: its edges; exclusive scan (n_mp + 1)
    CorbBAEdge* edges;
    int* status;
};
void bas_launch_vertices(const BAStoreDev& d, hipStream_t s);
void bas_launch_count(const BAStoreDev& d, hipStream_t s);
void bas_launch_fill(const BAStoreDev& d, hipStream_t s);
void bas_launch_writeback(const BAStoreDev& d, unsigned long long loop_kf, float scale_factor, hipStream_t s);
// after Optimizer::LocalBundleAdjustment: vToErase applied to the records, estimates written back, MapPoint::UpdateNormalAndDepth (store_kernels.hip)
void bas_launch_local_finish(const BAStoreDev& d, const uint8_t* edge_outlier, int apply_erase, float scale_factor, hipStream_t s);
// the call's host-bound results as ONE block of 32-bit words (one copy instead of four): [0] = outlier observations; from word pairs_off on (index into kf_slots, index into
// mp_slots) of each in edge order; from poses_off on the poses (n_kf x 16 floats); from points_off on the points (n_mp x 3 floats).  One workgroup.
void bas_launch_local_results(const BAStoreDev& d, const uint8_t* edge_outlier, int n_edges, int* block, int pairs_off, int poses_off, int points_off, hipStream_t s);
