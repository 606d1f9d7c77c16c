// store_internal.h -- record layouts of the device-resident keyframe / map-point stores (corb_store.cpp, corb_comm.cpp, map_kernels.hip)
#pragma once
#include "corb_internal.h"

// ---- keyframe record (what the reference serialises per KeyFrame, C/include/KeyFrame.h:59-87) ----
// F = max_features; every section 64-byte aligned:
//   header 256 B : KfHeader below (counts, mnId, mnClientId, flags, fx..mbf, Tcw, mTcwGBA, mnBAGlobalForKF, mvInvLevelSigma2)
//   kp[F] 28 B | desc[F] 32 B | u_right[F] f32 | depth[F] f32 | angle[F] f32 | flags[F] u8 | fv_node[F] u32 | fv_off[F+1] i32 | fv_idx[F] u32 | mp_id[F] u64
#define CORB_KF_HEADER_BYTES 256
struct KfHeader {
    int32_t n, n_nodes;                 // features, FeatureVector nodes
    CorbKeyFrameMeta m;                 // mnId, mnClientId, flags, fx..mbf, nlevels, Tcw, mTcwGBA, mnBAGlobalForKF, mvInvLevelSigma2 (include/corb_accel.h)
};
static_assert(sizeof(KfHeader) <= CORB_KF_HEADER_BYTES, "keyframe header does not fit");
static_assert(sizeof(KfHeader) == sizeof(CorbKeyFrameMeta) + 8, "KfHeader = {n, n_nodes} + CorbKeyFrameMeta");

struct RecLayout {
    size_t kp, desc, ur, depth, angle, flags, fv_node, fv_off, fv_idx, mp_id, bytes;
    __host__ __device__ explicit RecLayout(int F) {
        size_t o = CORB_KF_HEADER_BYTES;
        kp = o; o = al(o + (size_t)F * 28); desc = o; o = al(o + (size_t)F * 32); ur = o; o = al(o + (size_t)F * 4); depth = o; o = al(o + (size_t)F * 4);
        angle = o; o = al(o + (size_t)F * 4); flags = o; o = al(o + (size_t)F); fv_node = o; o = al(o + (size_t)F * 4); fv_off = o; o = al(o + ((size_t)F + 1) * 4);
        fv_idx = o; o = al(o + (size_t)F * 4); mp_id = o; o = al(o + (size_t)F * 8); bytes = o;
    }
    __host__ __device__ static size_t al(size_t v) { return (v + 63) & ~(size_t)63; }
};

// ---- map-point record (C/include/MapPoint.h:52-72) ----
//   header 128 B : CorbMapPointRecord (mnId, mpRefKF, mDescriptor, mnClientId, nObs, flags, mWorldPos, mNormalVector, mfMin/MaxDistance, mPosGBA, mnBAGlobalForKF)
//   obs_kf[O] u64 | obs_idx[O] u32     (mObservations: std::map<LightKeyFrame, size_t>, ascending keyframe id)
#define CORB_MP_HEADER_BYTES 128
static_assert(sizeof(CorbMapPointRecord) <= CORB_MP_HEADER_BYTES, "map point header does not fit");
static_assert(offsetof(CorbMapPointRecord, descriptor) % 8 == 0 && sizeof(CorbMapPointRecord) == 112, "the kernels read mDescriptor as aligned 64-bit words; 112 bytes is the wire format");
struct MpLayout {
    size_t obs_kf, obs_idx, scratch, bytes;     // scratch: CorbMapPointScratch behind the observation lists (round 6)
    __host__ __device__ explicit MpLayout(int O) { obs_kf = CORB_MP_HEADER_BYTES; obs_idx = obs_kf + (size_t)O * 8; scratch = RecLayout::al(obs_idx + (size_t)O * 4); bytes = RecLayout::al(scratch + sizeof(CorbMapPointScratch)); }
};
static_assert(sizeof(CorbMapPointScratch) == 104, "CorbMapPointScratch layout");

// slot record <- one keyframe's keypoints / descriptors / mvuRight / mvDepth (count read on the device when n_host < 0); clears flags, BoW groups, map-point ids
void corb_launch_kf_pack(const CorbKeyPoint* kp, const uint8_t* desc, const float* ur, const float* depth, const int* count, int n_host, unsigned long long id,
                         char* rec, int F, hipStream_t s);
// records[first .. first+n) <- host-side headers + CSR observation lists (already on the device)
void corb_launch_mp_pack(const CorbMapPointRecord* hdr, const int* obs_off, const unsigned long long* obs_kf, const uint32_t* obs_idx, int n, char* base, int first, int O, int* status, hipStream_t s);
// records -> headers + padded observation arrays [n][O]
void corb_launch_mp_unpack(const char* base, int first, int n, int O, CorbMapPointRecord* hdr, unsigned long long* obs_kf, uint32_t* obs_idx, hipStream_t s);
// dst[i] <- record slots[i] (contiguous staging of the records a push sends)
void corb_launch_gather_records(const char* base, size_t rec_bytes, const int* slots, int n, char* dst, hipStream_t s);
// MapFusion::insertServerMapToGlobleMap on records: Tcw <- Tcw * To2n for the keyframe slots, p <- Rwc (p - tcw) for the map-point slots
void corb_launch_rebase_records(const float* To2n, char* kf_base, size_t kf_bytes, const int* kf_slots, int n_kf, char* mp_base, size_t mp_bytes, const int* mp_slots, int n_mp, hipStream_t s);
// records[first .. first+n) <- whole keyframes from flat device arrays (CSR over the keyframes' features); NULL optional arrays as in corb_kf_store_put_batch
void corb_launch_kf_pack_batch(const CorbKeyFrameMeta* meta, const int* feat_off, const CorbKeyPoint* kp, const uint8_t* desc, const float* ur, const float* depth,
                               const unsigned long long* mp_id, int n, char* base, int first, int F, hipStream_t s);
