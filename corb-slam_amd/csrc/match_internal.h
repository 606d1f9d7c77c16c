// match_internal.h -- device-side argument blocks of the BoW / triangulation matchers.
#pragma once
#include "corb_internal.h"

struct CorbBowDev {
    int variant, check_ori, n_pairs;
    float nnratio;
    const int* pair_a; const int* pair_b;          // common vocabulary nodes: (node index in fv1, in fv2)
    const int* off1; const int* idx1;              // fv1 CSR
    const int* off2; const int* idx2;              // fv2 CSR
    const unsigned long long* desc1; const unsigned long long* desc2;
    const float* angle1; const float* angle2;
    const uint8_t* valid1; const uint8_t* valid2;
    int* match;                                    // variant 0: [n2] <- idx1 ; variant 1: [n1] <- idx2
    int* bin;                                      // rotation-histogram bin per output slot (-1 = none)
    int* hist;                                     // [CORB_HISTO_LENGTH]
    int* n_matches;
};

struct CorbTriDev {
    int n_queries, only_stereo, check_ori;         // n_queries: the number of queries, or (n_queries_dev != nullptr) an upper bound for the launch
    const int* n_queries_dev;                      // query count in device memory (queries built on the device: corb_launch_tri_queries)
    const int* q_idx1; const int* q_node2;         // per query: KF1 feature, node index in fv2
    const int* off2; const int* idx2;
    const unsigned long long* desc1; const unsigned long long* desc2;
    const CorbKeyPoint* kp1; const CorbKeyPoint* kp2;
    const float* uright1; const float* uright2;
    const uint8_t* has_mp2;
    float F12[9]; float ex, ey;
    const float* scale2; const float* sigma2_2;
    int* match; int* bin; int* hist; int* n_matches;
};

void corb_launch_bow(const CorbBowDev& d, int n_slots, hipStream_t stream);
void corb_launch_tri(const CorbTriDev& d, int n1, hipStream_t stream);
// queries of SearchForTriangulation built on the device: the features of the common vocabulary nodes (pa[k] in fv1, pb[k] in fv2) of KF1 that have no MapPoint
// (and a stereo observation if only_stereo), in any order; *counter (zeroed by the caller) receives their number
void corb_launch_tri_queries(const int* off1, const int* idx1, const uint8_t* flags1, const float* uright1, const int* pa, const int* pb, int n_common,
                             int only_stereo, int* q_idx1, int* q_node2, int* counter, hipStream_t stream);
void corb_launch_hamming_pairs(const uint8_t* a, const uint8_t* b, int n, int* out, hipStream_t stream);
