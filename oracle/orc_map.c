/* CPU ORACLE (test infrastructure) -- map maintenance arithmetic next to the hot path:
 *   MapPoint::ComputeDistinctiveDescriptors  (corbslam_client/src/MapPoint.cc:337-402)  N x N Hamming + row medians
 *   MapFusion::insertServerMapToGlobleMap    (corbslam_server/src/MapFusion.cpp:622-658) rigid re-basing of a client map
 *   MapPoint::Replace                        (corbslam_client/src/MapPoint.cc:277-316)   observation lists re-linked (list arithmetic only)
 * cv::Mat products on CV_32F = cv::gemm: double accumulation, one rounding to float.  See orc.h for scope. */
#include "orc.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

/* per map point p: descriptors desc[offset[p] .. offset[p+1]) (one per non-bad observing keyframe, in std::map order);
 * best_idx[p] = row with the least median distance to the rest (first such row), -1 for a point without descriptors */
void orc_distinctive_descriptors(const uint8_t* desc, const int32_t* offset, int n_points, int32_t* best_idx)
{
    for (int p = 0; p < n_points; p++) {
        const int N = offset[p + 1] - offset[p];
        best_idx[p] = -1;
        if (N <= 0) continue;
        const uint8_t* D = desc + (size_t)offset[p] * 32;
        int* row = (int*)malloc(sizeof(int) * N);
        int BestMedian = INT_MAX, BestIdx = 0;
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < N; j++) row[j] = i == j ? 0 : orc_descriptor_distance(D + (size_t)i * 32, D + (size_t)j * 32);
            qsort(row, N, sizeof(int), cmp_int);
            const int median = row[(int)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best_idx[p] = BestIdx;
        free(row);
    }
}

/* Tcw <- Tcw * To2n for every keyframe; p <- Rwc * (p - tcw) with tcw = To2n(0:3,3), Rwc = To2n(0:3,0:3)^T for every map point */
void orc_rebase_map(const float* To2n, float* poses, int n_poses, float* points, int n_points)
{
    for (int k = 0; k < n_poses; k++) {
        float* T = poses + 16 * (size_t)k; float o[16];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
            double s = 0; for (int c = 0; c < 4; c++) s += (double)T[i * 4 + c] * (double)To2n[c * 4 + j];
            o[i * 4 + j] = (float)s;
        }
        memcpy(T, o, sizeof(o));
    }
    for (int m = 0; m < n_points; m++) {
        float* p = points + 3 * (size_t)m;
        const float d[3] = { p[0] - To2n[3], p[1] - To2n[7], p[2] - To2n[11] };
        float o[3];
        for (int i = 0; i < 3; i++) { double s = 0; for (int c = 0; c < 3; c++) s += (double)To2n[c * 4 + i] * (double)d[c]; o[i] = (float)s; }
        p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
    }
}

/* void MapPoint::Replace(MapPoint* pMP) (MapPoint.cc:277-316) on flat observation lists.  `this` is observed by (kf_this[k], idx_this[k]), k < n_this, in the
 * order of its std::map<LightKeyFrame, size_t> (ascending keyframe id, MapPoint.h:182); pMP by (kf_into[j], idx_into[j]), j < *n_into, room for cap_into entries.
 * Per observation of `this`, in map order (:300-313):
 *   action[k] = 1: pMP is not in that keyframe (IsInKeyFrame, :304) -> pKF->ReplaceMapPointMatch(idx, pMP) and pMP->AddObservation(pKF, idx) (:305-306; AddObservation
 *                  files mObservations[pKF] = idx, MapPoint.cc:150-153: the entry lands at its place in the ascending list);
 *   action[k] = 2: pMP is in that keyframe already -> pKF->EraseMapPointMatch(idx) (:309).
 * counters = {mnVisible, mnFound}: pMP->IncreaseFound(nfound); pMP->IncreaseVisible(nvisible) (:312-313).  `this` ends without observations, bad, mpReplaced = pMP
 * (:288-291: the caller's part, nothing to compute).  Returns 1 if both are the same point (:278-279, nothing done), -1 if pMP's list has no room (the records' fixed
 * capacity; nothing done), else 0. */
int orc_mappoint_replace(uint64_t id_this, uint64_t id_into, const uint64_t* kf_this, const uint32_t* idx_this, int n_this,
                         uint64_t* kf_into, uint32_t* idx_into, int32_t* n_into, int cap_into, uint8_t* action, const int32_t* counters_this, int32_t* counters_into)
{
    if (id_this == id_into) return 1;
    int nb = *n_into, fresh = 0;
    for (int k = 0; k < n_this; k++) { int in = 0; for (int j = 0; j < nb; j++) in |= kf_into[j] == kf_this[k]; fresh += !in; }
    if (nb + fresh > cap_into) return -1;
    for (int k = 0; k < n_this; k++) {
        int in = 0;
        for (int j = 0; j < nb; j++) in |= kf_into[j] == kf_this[k];       /* pMP->IsInKeyFrame(pKF): evaluated on the list as the entries before k left it */
        if (!in) {
            int j = nb;
            while (j > 0 && kf_into[j - 1] > kf_this[k]) { kf_into[j] = kf_into[j - 1]; idx_into[j] = idx_into[j - 1]; j--; }
            kf_into[j] = kf_this[k]; idx_into[j] = idx_this[k]; nb++;
            action[k] = 1;
        } else action[k] = 2;
    }
    *n_into = nb;
    counters_into[0] += counters_this[0]; counters_into[1] += counters_this[1];
    return 0;
}
