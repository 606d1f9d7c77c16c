// store_internal.h -- record layout of the device-resident keyframe store (corb_store.cpp, map_kernels.hip)
#pragma once
#include "corb_internal.h"

// F = max_features; every section 64-byte aligned:
//   header 64 B : int32 n, int32 n_nodes, uint64 keyframe id
//   kp[F] 28 B | desc[F] 32 B | u_right[F] f32 | depth[F] f32 | angle[F] f32 | flags[F] u8 | fv_node[F] u32 | fv_off[F+1] i32 | fv_idx[F] u32
struct RecLayout {
    size_t kp, desc, ur, depth, angle, flags, fv_node, fv_off, fv_idx, bytes;
    __host__ __device__ explicit RecLayout(int F) {
        size_t o = 64;
        kp = o; o = al(o + (size_t)F * 28); desc = o; o = al(o + (size_t)F * 32); ur = o; o = al(o + (size_t)F * 4); depth = o; o = al(o + (size_t)F * 4);
        angle = o; o = al(o + (size_t)F * 4); flags = o; o = al(o + (size_t)F); fv_node = o; o = al(o + (size_t)F * 4); fv_off = o; o = al(o + ((size_t)F + 1) * 4);
        fv_idx = o; o = al(o + (size_t)F * 4); bytes = o;
    }
    __host__ __device__ static size_t al(size_t v) { return (v + 63) & ~(size_t)63; }
};

// slot record <- one keyframe's keypoints / descriptors / mvuRight / mvDepth (count read on the device when n_host < 0); clears flags and BoW groups
void corb_launch_kf_pack(const CorbKeyPoint* kp, const uint8_t* desc, const float* ur, const float* depth, const int* count, int n_host, unsigned long long id,
                         char* rec, int F, hipStream_t s);
