// lane_exchange.h -- cross-lane exchanges of a wavefront's reductions WITHOUT the LDS crossbar (gfx950).
// `__shfl_xor` of a double compiles to two ds_bpermute_b32 behind an address computation: an LDS round trip per butterfly stage, on the one LDS pipe all wavefronts of a
// compute unit share.  For the constant xor masks of a butterfly the data-parallel primitives do the same move in the vector ALU:
//   xor 1, 2 : DPP quad_perm            xor 4 : DPP row_shl:4 into banks 0, 2 + row_shr:4 into banks 1, 3            xor 8 : DPP row_ror:8
//   xor 16   : v_permlane16_swap        xor 32 : v_permlane32_swap   (swap the odd rows / upper half of one register with the even rows / lower half of another:
//              called on (v, v) the two results hold {own, partner} on one half of the lanes and {partner, own} on the other -- any COMMUTATIVE combination of the
//              two is "own op partner" on every lane)
// Sums are own + partner or partner + own: the same bits as the __shfl_xor forms (tools/ubench/lane_exchange.hip checks every move against __shfl_xor on the device).
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned lx_v2u __attribute__((ext_vector_type(2)));
template <int CTRL, int BANK> __device__ __forceinline__ double lx_dpp(double old, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}
// the value of lane (l ^ M), M = 1, 2, 4, 8
template <int M> __device__ __forceinline__ double lx_xor(double v)
{
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "DPP moves stay inside a row of 16 lanes");
    if (M == 1) return lx_dpp<0xB1, 0xF>(v, v);
    if (M == 2) return lx_dpp<0x4E, 0xF>(v, v);
    if (M == 8) return lx_dpp<0x128, 0xF>(v, v);
    return lx_dpp<0x114, 0xA>(lx_dpp<0x104, 0x5>(v, v), v);
}
// lanes without bit 5 (bit 4): a + the partner's a; lanes with it: b + the partner's b   (the transposing butterfly's step; a == b: v + partner's v on every lane)
__device__ __forceinline__ double lx_xadd32(double a, double b)
{
    const lx_v2u r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const lx_v2u r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)r1.x, (int)r0.x) + __hiloint2double((int)r1.y, (int)r0.y);
}
__device__ __forceinline__ double lx_xadd16(double a, double b)
{
    const lx_v2u r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const lx_v2u r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)r1.x, (int)r0.x) + __hiloint2double((int)r1.y, (int)r0.y);
}
// v + the value of lane (l ^ M), any power of two below 64
template <int M> __device__ __forceinline__ double lx_add_xor(double v)
{
    if (M == 32) return lx_xadd32(v, v);
    if (M == 16) return lx_xadd16(v, v);
    return v + lx_xor<(M < 16 ? M : 1)>(v);
}
// max(v, the value of lane (l ^ M))
template <int M> __device__ __forceinline__ double lx_max_xor(double v)
{
    if (M >= 16) {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const lx_v2u r0 = M == 32 ? __builtin_amdgcn_permlane32_swap(lo, lo, false, false) : __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const lx_v2u r1 = M == 32 ? __builtin_amdgcn_permlane32_swap(hi, hi, false, false) : __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        return fmax(__hiloint2double((int)r1.x, (int)r0.x), __hiloint2double((int)r1.y, (int)r0.y));
    }
    return fmax(v, lx_xor<(M < 16 ? M : 1)>(v));
}
// the butterfly sum over the 64 lanes in the order 32, 16, 8, 4, 2, 1 (what `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o)` computes, bit for bit)
__device__ __forceinline__ double lx_wave_sum(double v)
{
    v = lx_xadd32(v, v); v = lx_xadd16(v, v);
    v += lx_xor<8>(v); v += lx_xor<4>(v); v += lx_xor<2>(v); v += lx_xor<1>(v);
    return v;
}
__device__ __forceinline__ double lx_wave_max(double v)
{
    v = lx_max_xor<32>(v); v = lx_max_xor<16>(v); v = lx_max_xor<8>(v); v = lx_max_xor<4>(v); v = lx_max_xor<2>(v); v = lx_max_xor<1>(v);
    return v;
}
